"""Shared test inputs: the fixtures the reference's own tests ship (copied as data under tests/golden/ref_fixtures/),
seeded synthetic models/graphs, and thin wrappers that run the ORACLE on them.  Nothing here reads /root/reference."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import yaml

from montreal_forced_aligner_amd import graph as G
from montreal_forced_aligner_amd import kaldi_io as K
from montreal_forced_aligner_amd import model as M
from oracle import oracle as O

GOLDEN = Path(__file__).resolve().parent / "golden"
REF = GOLDEN / "ref_fixtures"


class Fixtures:
    def __init__(self):
        ar = K.load_acoustic_model_archive(REF / "mono_model.zip")
        self.mono_meta = yaml.safe_load(ar["meta.yaml"])
        self.mono_tm, self.mono_am = M.load_model_bytes(ar["final.mdl"])
        self.mono_tree = K.read_tree(ar["tree"])
        ar2 = K.load_acoustic_model_archive(REF / "acoustic_g2p_output_model.zip")
        self.g2p_archive = ar2
        self.g2p_meta = json.loads(ar2["meta.json"])
        self.g2p_tm, self.g2p_am = M.load_model_bytes(ar2["final.mdl"])
        self.g2p_tree = K.read_tree(ar2["tree"])
        self.g2p_lda = K.read_matrix_file(ar2["lda.mat"])
        pcm, sr = K.read_wav_pcm16(REF / "acoustic_corpus.wav")
        assert sr == 16000
        self.pcm = pcm[0]
        self.text = (REF / "acoustic_corpus.lab").read_text().strip()
        # MFA 1.x/2.0.0 phone table of mono_model: silence phones sil, sp (optional silence), spn
        lex = G.LexiconCompiler(position_dependent_phones=True, phones=self.mono_meta["phones"], silence_phone="sp",
                                oov_phone="spn")
        lex.load_pronunciations(REF / "test_acoustic.txt")
        lex.build_phone_table(["sil", "sp", "spn"])
        self.mono_lex = lex
        self.mono_gc = G.TrainingGraphCompiler(self.mono_tm, self.mono_tree, lex)

    def mono_graph(self, text, transition_scale=1.0, self_loop_scale=0.1):
        f = self.mono_gc.compile_fst(text)
        return G.add_transition_probs(f, self.mono_tm.scaled_log_probs(transition_scale, self_loop_scale))

    def mono_feats(self, wave_i16, snip_edges=0):
        mf = O.mfcc(wave_i16.astype(np.float32), O.default_mfcc_opts(snip_edges=snip_edges))
        return O.deltas(O.cmvn_apply(O.cmvn_stats([mf]), mf))


def oracle_align(tm, fst, loglikes, pdf_list, acoustic_scale=0.1, beam=10.0, retry_beam=40.0, want_stats=False):
    """loglikes: [T, len(pdf_list)] (columns follow pdf_list)."""
    lut = np.zeros(tm.num_pdfs, dtype=np.int32)
    lut[np.asarray(pdf_list)] = np.arange(len(pdf_list), dtype=np.int32)
    tid2col = lut[np.maximum(tm.id2pdf, 0)].astype(np.int32)
    return O.align(fst.num_states, fst.start, fst.arc_offsets, fst.arcs, fst.final, loglikes, tid2col, acoustic_scale,
                   beam, retry_beam, want_stats=want_stats)


def oracle_align_feats(tm, fst, feats, am, acoustic_scale=0.1, beam=10.0, retry_beam=40.0):
    """The oracle with Kaldi's lazy decodable (features + model in, scores on demand): what GmmAligner.align_utterance runs."""
    return O.align_feats(fst.num_states, fst.start, fst.arc_offsets, fst.arcs, fst.final, feats, am.gconsts, am.means_invvars,
                         am.inv_vars, am.pdf_offsets, tm.id2pdf, acoustic_scale, beam, retry_beam)


def random_gmm(rng, dim, gauss_per_pdf):
    """Seeded diagonal GMM with the given number of Gaussians per pdf (Kaldi gconst convention)."""
    gconsts, mi, iv, offs = [], [], [], [0]
    for g in gauss_per_pdf:
        w = rng.dirichlet(np.ones(g)) if g > 1 else np.ones(1)
        mean = rng.normal(0, 3.0, size=(g, dim))
        var = rng.uniform(0.5, 4.0, size=(g, dim))
        inv = 1.0 / var
        gc = np.log(w) - 0.5 * (dim * np.log(2 * np.pi) + np.log(var).sum(axis=1) + (mean * mean * inv).sum(axis=1))
        gconsts.append(gc.astype(np.float32)); mi.append((mean * inv).astype(np.float32)); iv.append(inv.astype(np.float32))
        offs.append(offs[-1] + g)
    return M.DiagGmmModel(dim, np.concatenate(gconsts), np.concatenate(mi), np.concatenate(iv), np.asarray(offs, np.int32))


def device_status(ref, n_frames):
    """The status the device decoders report for an utterance the oracle aligned: the oracle's, except that a best path with
    more word labels than frames (output labels on epsilon arcs) cannot be returned in d_words — status 7 (include/mfa_hip.h)."""
    if ref["status"] in (0, 1) and len(ref["words"]) > n_frames:
        return 7
    return ref["status"]


def has_negative_eps_cycle(fst) -> bool:
    """A cycle of epsilon input arcs with negative total weight: Kaldi's ProcessNonemitting does not terminate on it (and the
    oracle, a faithful restatement, runs out of memory) — fuzzers must not generate one.  Bellman-Ford over the epsilon arcs."""
    arcs = fst.arcs
    eps = arcs["ilabel"] == 0
    if not eps.any():
        return False
    src = np.repeat(np.arange(fst.num_states), np.diff(fst.arc_offsets))[eps]
    dst = arcs["nextstate"][eps].astype(np.int64)
    w = arcs["weight"][eps].astype(np.float64)
    if not (w < 0).any():
        return False
    dist = np.zeros(fst.num_states, dtype=np.float64)
    for _ in range(fst.num_states + 1):
        cand = dist[src] + w
        new = dist.copy()
        np.minimum.at(new, dst, cand)
        if np.array_equal(new, dist):
            return False
        dist = new
    return True
