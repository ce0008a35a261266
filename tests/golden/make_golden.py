"""Generate tests/golden/oracle_vectors.npz — the golden vectors SURVEY §8(c) asks the build to create itself.

The reference's numerics live in kalpy/Kaldi, which is not installable here, and its own tests pin nothing at frame level
("parity unpinned"), so these vectors are the output of THIS repository's C++ oracle (oracle/mfa_oracle.cpp) on the
reference's own plumbing fixtures (tests/golden/ref_fixtures: mono_model.zip, acoustic_g2p_output_model.zip,
acoustic_corpus.wav, test_acoustic.txt), cross-checked at generation time against the independent numpy restatement
(oracle/np_oracle.py).  They freeze the oracle: a later change to the oracle or to the device path that moves any of these
numbers is caught by tests/test_golden_cpu.py / tests/test_gpu_golden.py.

    python tests/golden/make_golden.py        # rewrites oracle_vectors.npz next to this file
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import np_oracle as NP  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import helpers  # noqa: E402

SECONDS = 3.0
TEXT = "this is the acoustic corpus i'm talking"


def main():
    fx = helpers.Fixtures()
    pcm = fx.pcm[: int(16000 * SECONDS)]
    wave = pcm.astype(np.float32)
    out = {"pcm_samples": np.int64(len(pcm)), "text": np.array(TEXT)}
    for snip in (0, 1):
        mf = O.mfcc(wave, O.default_mfcc_opts(snip_edges=snip))
        ref = NP.mfcc(wave, snip_edges=snip)
        assert np.abs(mf - ref).max() < 2e-3, np.abs(mf - ref).max()     # two float32 FFTs on int16-scale audio
        out[f"mfcc_snip{snip}"] = mf
    mf = out["mfcc_snip0"]
    stats = O.cmvn_stats([mf])
    x = O.deltas(O.cmvn_apply(stats, mf))
    assert np.abs(x - NP.deltas(NP.cmvn(mf))).max() < 1e-4
    out["cmvn_stats"], out["delta_feats"] = stats, x
    tm, am = fx.mono_tm, fx.mono_am
    all_pdfs = np.arange(am.num_pdfs)
    ll = O.gmm_loglikes(x[:50], am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, all_pdfs)
    assert np.abs(ll - NP.gmm_loglikes(x[:50], am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, all_pdfs)).max() < 1e-3
    out["loglikes_first50_allpdfs"] = ll
    fst = fx.mono_graph(TEXT)
    out["graph_arc_offsets"], out["graph_final"] = fst.arc_offsets.astype(np.int64), fst.final.astype(np.float32)
    for k in ("ilabel", "olabel", "weight", "nextstate"):
        out[f"graph_{k}"] = np.ascontiguousarray(fst.arcs[k])
    out["graph_start"] = np.int32(fst.start)
    pl = np.unique(tm.id2pdf[fst.arcs["ilabel"]])
    lls = O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pl)
    res = helpers.oracle_align(tm, fst, lls, pl, beam=100.0, retry_beam=400.0)
    assert res["status"] in (0, 1)
    out["ali"], out["words"], out["like"], out["status"] = res["ali"], res["words"], np.float32(res["like"]), np.int32(res["status"])
    ph, ok = O.split_to_phones(res["ali"], tm.id2state, tm.is_self_loop, tm.is_final, tm.tuples)
    assert ok
    out["phone_intervals"] = np.asarray(ph, dtype=np.int32)          # rows as orc_split_to_phones returns them
    # splice ±3 + LDA with the g2p model's matrix (the [T,40] path)
    lda = fx.g2p_lda
    y = O.affine(O.splice(O.cmvn_apply(stats, mf)), lda)
    assert np.abs(y - NP.affine(NP.splice(NP.cmvn(mf)), lda)).max() < 1e-3
    out["lda_feats"] = y
    np.savez_compressed(Path(__file__).with_name("oracle_vectors.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
