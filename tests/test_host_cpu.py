"""CPU tests of the host side: the C ABI library loads and exports every symbol include/mfa_hip.h declares, the rank
sharding follows the reference's rule, and a world_size-2 gloo run shards and gathers with no data-path collective."""
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_c_abi_exports_every_declared_symbol():
    from montreal_forced_aligner_amd import _lib

    _lib.build_native()
    header = (ROOT / "include" / "mfa_hip.h").read_text()
    declared = set(re.findall(r"MFA_API\s+[\w\s\*]+?\b(mfa_\w+)\s*\(", header))
    assert len(declared) >= 20
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.lib()  # loads without a GPU (no compute calls here)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mfa_version() >= 1
    out = subprocess.check_output(["nm", "-D", "--defined-only", str(ROOT / "montreal_forced_aligner_amd" / "libmfa_hip.so")]).decode()
    exported = set(re.findall(r"\bT (mfa_\w+)", out))
    assert declared <= exported


def test_fst_first_frames_host_helper(fx):
    """mfa_fst_first_frames is pure host code (a breadth-first search): state depth = first frame a token can reach it."""
    import ctypes as C

    from montreal_forced_aligner_amd import _lib

    lib = _lib.lib()
    f = fx.mono_graph("this is the acoustic corpus")
    arc_off = np.ascontiguousarray(f.arc_offsets, dtype=np.int32)
    nxt = np.ascontiguousarray(f.arcs["nextstate"], dtype=np.int32)
    depth = np.empty(f.num_states, dtype=np.int32)
    assert lib.mfa_fst_first_frames(f.num_states, arc_off.ctypes.data, nxt.ctypes.data, int(f.start), depth.ctypes.data) == 0
    ref = np.full(f.num_states, np.iinfo(np.int32).max, dtype=np.int64)
    ref[f.start] = 0
    src = np.repeat(np.arange(f.num_states), np.diff(f.arc_offsets))
    for _ in range(f.num_states):
        nd = ref.copy()
        ok = ref[src] < np.iinfo(np.int32).max
        np.minimum.at(nd, nxt[ok], ref[src][ok] + 1)
        if np.array_equal(nd, ref):
            break
        ref = nd
    assert np.array_equal(depth.astype(np.int64), ref)
    assert depth[f.start] == 0 and depth.max() < f.num_states  # trimmed graph: everything reachable
    # malformed input is rejected, not walked
    assert lib.mfa_fst_first_frames(f.num_states, arc_off.ctypes.data, nxt.ctypes.data, f.num_states, depth.ctypes.data) != 0


def test_engine_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from montreal_forced_aligner_amd import _lib
    from montreal_forced_aligner_amd.engine import AlignmentEngine

    with pytest.raises(_lib.MfaHipError):
        AlignmentEngine(0)


def test_product_package_never_imports_the_oracle():
    pkg = ROOT / "montreal_forced_aligner_amd"
    for py in pkg.rglob("*.py"):
        src = py.read_text()
        assert "oracle" not in re.sub(r"#.*", "", src).replace("the oracle", "").replace("oracle's", ""), py


def test_speaker_sharding_follows_reference_rule():
    from montreal_forced_aligner_amd import sharding

    utt2spk = np.array([0] * 5 + [1] * 3 + [2] * 3 + [3] * 1 + [4] * 1)
    r = sharding.assign_speakers(utt2spk, 2)
    # speakers by count 5,3,3,1,1 → job0:5, job1:3, job1:+3=6, job0:+1=6, job0 (tie → lowest id):+1
    assert [int(r[utt2spk == s][0]) for s in range(5)] == [0, 1, 1, 0, 0]
    for s in range(5):
        assert len(set(r[utt2spk == s])) == 1  # a speaker never straddles ranks (CMVN/fMLLR are per speaker)
    c = sharding.assign_contiguous(10, 3)
    assert c.tolist() == [0, 0, 0, 1, 1, 1, 2, 2, 2, 2]
    w = sharding.assign_speakers(utt2spk, 2, weights=np.where(utt2spk == 3, 100.0, 1.0))
    assert int(w[utt2spk == 3][0]) == 0 and int(w[utt2spk == 0][0]) == 1


_WORKER = r"""
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["REPO_ROOT"])
from montreal_forced_aligner_amd import sharding
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["PORT"], rank=int(os.environ["RANK"]), world_size=2)
rank = dist.get_rank()
rng = np.random.default_rng(0)
utt2spk = rng.integers(0, 7, size=40)
ranks = sharding.assign_speakers(utt2spk, 2)
mine = sharding.local_indices(ranks, rank)
local = {int(u): ("ali-of-%d" % u, rank) for u in mine}      # stand-in for per-utterance alignments
dist.barrier()
allres = sharding.gather_results(local, 2)
assert sorted(allres) == list(range(40)), sorted(allres)
for u, (tag, r) in allres.items():
    assert tag == "ali-of-%d" % u and r == ranks[u]
print("rank", rank, "ok", len(mine))
dist.destroy_process_group()
"""


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), PORT=port, REPO_ROOT=str(ROOT))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert all("ok" in o for o in outs)


def test_compressed_matrix_round_trip_and_scp(tmp_path):
    """Kaldi CompressedMatrix ("CM") writer against the reader: values come back within the format's quantisation step
    (1/64, 1/128, 1/63 of the column's inter-percentile segments), headers are monotone, scp offsets land on the objects."""
    import io

    from montreal_forced_aligner_amd import kaldi_io as K

    rng = np.random.default_rng(5)
    m = (rng.standard_normal((137, 13)) * np.linspace(0.5, 30, 13) + np.linspace(-20, 40, 13)).astype(np.float32)
    m[:, 3] = 7.25                      # a constant column still needs strictly increasing percentiles
    buf = io.BytesIO()
    K.write_compressed_matrix(buf, m)
    assert len(buf.getvalue()) == 3 + 16 + 8 * 13 + 137 * 13       # "CM " + global header + column headers + one byte per value
    back = K.BinaryReader(buf.getvalue()).matrix()
    assert back.shape == m.shape
    for c in range(m.shape[1]):
        sd = np.sort(m[:, c])
        q = m.shape[0] // 4
        step = max(sd[q] - sd[0], sd[3 * q] - sd[q], sd[-1] - sd[3 * q]) / 63.0 + (m.max() - m.min()) / 65535.0 * 2
        assert np.abs(back[:, c] - m[:, c]).max() <= step, c
    small = rng.standard_normal((3, 4)).astype(np.float32)      # fewer than five rows: Kaldi's small-matrix rule
    buf = io.BytesIO()
    K.write_compressed_matrix(buf, small)
    assert np.abs(K.BinaryReader(buf.getvalue()).matrix() - small).max() < 0.05 * (small.max() - small.min()) + 1e-3
    # table + scp
    ark, scp = tmp_path / "feats.ark", tmp_path / "feats.scp"
    K.write_table(ark, [("spk1-utt1", m), ("spk1-utt2", small)], "compressed_matrix", scp)
    entries = K.read_scp(scp)
    assert [e[0] for e in entries] == ["spk1-utt1", "spk1-utt2"]
    cache = {}
    assert K.read_scp_object(cache, entries[1][1], entries[1][2], "matrix").shape == (3, 4)
    assert [k for k, _ in K.read_ark(ark.read_bytes(), "matrix")] == ["spk1-utt1", "spk1-utt2"]


def _parse_text_fst(text):
    arcs, finals = [], {}
    for line in text.strip().split("\n"):
        f = line.split()
        if len(f) <= 2:
            finals[int(f[0])] = float(f[1]) if len(f) == 2 else 0.0
        else:
            arcs.append((int(f[0]), int(f[1]), f[2], f[3], float(f[4]) if len(f) > 4 else 0.0))
    n = 1 + max([a[0] for a in arcs] + [a[1] for a in arcs] + list(finals))
    return n, arcs, finals


def test_lexicon_fst_equals_the_references_expected_file():
    """The only vector the reference holds for the graph builder: tests/data/dictionaries/expected/
    {lexicon.text.fst, phones.txt, words.txt} for a two-word position-dependent dictionary (copied as data under
    tests/golden/ref_fixtures/expected_*).  LexiconCompiler's phone table must equal phones.txt line for line and its
    L.fst must equal lexicon.text.fst up to a renumbering of states, weights to 1e-12."""
    import itertools

    from montreal_forced_aligner_amd import graph as G
    from tests import helpers

    lex = G.LexiconCompiler(position_dependent_phones=True, silence_word="!SIL", oov_word="<unk>", silence_phone="sil",
                            oov_phone="spn", silence_probability=0.5, initial_silence_probability=0.5, ignore_case=False)
    for w, p in (("!SIL", "sil"), ("<unk>", "spn"), ("worda", "phonea phoneb"), ("wordb", "phonea phonec")):
        lex.add_pronunciation(G.Pronunciation(w, p))
    lex.build_phone_table()
    want_phones = [ln.split() for ln in (helpers.REF / "expected_phones.txt").read_text().strip().split("\n")]
    assert [[s, str(k)] for k, s in lex.phone_table] == want_phones
    want_words = [ln.split()[0] for ln in (helpers.REF / "expected_words.txt").read_text().strip().split("\n")]
    got_words = [s for _k, s in lex.word_table]
    assert want_words[0] == "<eps>" and want_words[1:5] == got_words[:4]      # same order after the reference's <eps> entry
    na, arcs_a, fin_a = _parse_text_fst(lex.lexicon_fst_text())
    nb, arcs_b, fin_b = _parse_text_fst((helpers.REF / "expected_lexicon.text.fst").read_text())
    assert na == nb and len(arcs_a) == len(arcs_b)

    def canon(arcs, finals, perm):
        return (sorted((perm[s], perm[d], il, ol, round(w, 12)) for s, d, il, ol, w in arcs),
                sorted((perm[s], round(w, 12)) for s, w in finals.items()))

    target = canon(arcs_b, fin_b, list(range(nb)))
    assert any(canon(arcs_a, fin_a, (0,) + perm) == target for perm in itertools.permutations(range(1, na))), \
        "no renumbering of states maps the generated L.fst onto the reference's expected one"


def test_suffix_sharing_keeps_every_path_of_the_phone_graph(fx):
    """LexiconCompiler(share_suffixes=True) (default) merges states with identical futures, as the reference's
    MinimizeEncoded step does (kalpy TrainingGraphCompiler → Kaldi training-graph-compiler.cc).  The multiset of
    (phone sequence, word sequence, cost) paths must not change: enumerated exhaustively on short transcripts, and on
    the long one the unpruned best path through the compiled graph keeps its cost and its labels."""
    import copy

    from montreal_forced_aligner_amd import graph as G
    from oracle import oracle as O

    lex = fx.mono_lex
    plain = copy.copy(lex)
    plain.share_suffixes = False
    assert lex.share_suffixes

    def paths(pg):
        out = []

        def walk(u, phones, words, cost):
            if u in pg.final:
                out.append((tuple(phones), tuple(words), round(cost + pg.final[u], 9)))
            for (v, ph, ol, w) in pg.arcs[u]:
                walk(v, phones + [ph], words + ([ol] if ol else []), cost + w)

        walk(pg.start, [], [], 0.0)
        return sorted(out)

    for text in ("this is", "the acoustic corpus", "i'm"):
        a, b = plain.phone_graph(text.split()), lex.phone_graph(text.split())
        pa, pb = paths(a), paths(b)
        assert sorted(set(pa)) == sorted(set(pb)) and len(pb) == len(set(pb))
        assert len(b.arcs) < len(a.arcs)
    text = " ".join(fx.text.split()[:12])
    tm = fx.mono_tm
    scaled = tm.scaled_log_probs(1.0, 0.1)
    # (determinize=False: the phone-level sharing on its own — DeterminizeStar + MinimizeEncoded would find the copies too)
    gc_plain = G.TrainingGraphCompiler(tm, fx.mono_tree, plain, determinize=False)
    f_a = G.add_transition_probs(gc_plain.compile_fst(text), scaled)
    f_b = G.add_transition_probs(G.TrainingGraphCompiler(tm, fx.mono_tree, lex, determinize=False).compile_fst(text), scaled)
    assert f_b.num_states < 0.7 * f_a.num_states
    x = fx.mono_feats(fx.pcm[: 16000 * 5])
    am = fx.mono_am
    pl = np.arange(am.num_pdfs)
    ll = O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pl)
    from tests import helpers
    r_a = helpers.oracle_align(tm, f_a, ll, pl, beam=1e9, retry_beam=0.0)
    r_b = helpers.oracle_align(tm, f_b, ll, pl, beam=1e9, retry_beam=0.0)
    assert r_a["status"] == 0 and r_b["status"] == 0
    assert np.array_equal(r_a["words"], r_b["words"]) and np.array_equal(r_a["ali"], r_b["ali"])
    assert abs(r_a["like"] - r_b["like"]) < 1e-2


def test_export_graphs_takes_the_references_utterance_records(fx, tmp_path):
    """CompileTrainGraphsFunction hands export_graphs the utterance table's dicts (kaldi_id + normalized_text,
    MFA/alignment/multiprocessing.py:547-571) and an empty interjection-cost dict for alignment workflows; the table written is
    the one (key, text) pairs give, graph for graph.  Non-empty interjection costs (transcript verification) and use_g2p are
    refused, not ignored."""
    import pytest

    from montreal_forced_aligner_amd import kaldi_io as K
    from montreal_forced_aligner_amd import kalpy_api as KA

    gc = KA.TrainingGraphCompiler(fx.mono_tm, fx.mono_tree, fx.mono_lex)
    recs = [{"kaldi_id": "1-1", "normalized_text": "this is the acoustic corpus", "text": "ignored"},
            {"kaldi_id": "1-2", "normalized_text": "um and that should be all thanks", "text": "ignored"}]
    gc.export_graphs(tmp_path / "a.ark", recs, interjection_words={})
    gc.export_graphs(tmp_path / "b.ark", [(r["kaldi_id"], r["normalized_text"]) for r in recs])
    assert (tmp_path / "a.ark").read_bytes() == (tmp_path / "b.ark").read_bytes()
    got = list(K.read_ark((tmp_path / "a.ark").read_bytes(), "fst"))
    assert [k for k, _ in got] == ["1-1", "1-2"]
    arch = KA.FstArchive(tmp_path / "a.ark")                   # what AlignFunction opens on the table (multiprocessing.py:828)
    assert [k for k, _ in arch] == ["1-1", "1-2"] and arch["1-2"].num_arcs == got[1][1].num_arcs
    with pytest.raises(KeyError):
        arch["1-3"]
    ref = fx.mono_gc.compile_fst(recs[1]["normalized_text"])
    assert got[1][1].num_states == ref.num_states and np.array_equal(got[1][1].arcs, ref.arcs)
    with pytest.raises(NotImplementedError):
        gc.export_graphs(tmp_path / "c.ark", recs, interjection_words={"uh": 1.0})
    with pytest.raises(NotImplementedError):
        KA.TrainingGraphCompiler(fx.mono_tm, fx.mono_tree, fx.mono_lex, use_g2p=True)


def test_determinize_star_log_and_minimize_encoded(fx):
    """Kaldi's TrainingGraphCompiler determinises (DeterminizeStarInLog) and minimises (MinimizeEncoded, weights quantised to
    1/1024) the HMM-level graph before the self-loops go in; graph.determinize_star_log / minimize_encoded restate the two.
    Checked here on what makes them what they are: (a) a toy graph — shared input prefixes merge, the shared arc carries the
    log-semiring sum of the alternatives, the alternatives the remainders, paths with identical labels are ⊕-added; (b) the
    compiled graphs of the reference's dictionary — deterministic, smaller, every (transition-id sequence, word sequence)
    of the undeterminised graph still there with its cost up to the quantum per arc; (c) alignment through both graphs: same
    transition-ids, likelihood equal up to the quantisation."""
    import math

    from montreal_forced_aligner_amd import graph as G
    from oracle import oracle as O
    from tests import helpers

    # (a) 0 -a/1-> 1 -b/0.5-> 3(final) ; 0 -a/2-> 2 -c/0-> 3 ; and a duplicate of the first path with other weights
    arcs = [[(1, 7, 5, 1.0), (2, 7, 5, 2.0), (4, 7, 5, 3.0)], [(3, 8, 0, 0.5)], [(3, 9, 0, 0.0)], [], [(3, 8, 0, 0.25)]]
    det = G.determinize_star_log(arcs, {3: 0.0}, 0)
    assert det is not None
    d_arcs, d_final = det
    ladd = lambda a, b: -math.log(math.exp(-a) + math.exp(-b))          # noqa: E731
    assert len(d_arcs[0]) == 1 and d_arcs[0][0][1:3] == (7, 5)           # one arc for label 7, the word label agreed at once
    assert abs(d_arcs[0][0][3] - ladd(ladd(1.0, 2.0), 3.0)) < 1e-12
    nxt = d_arcs[d_arcs[0][0][0]]
    assert [a[1] for a in nxt] == [8, 9]                                  # by ascending input label
    total_b = ladd(1.0 + 0.5, 3.0 + 0.25)                                 # the two paths spelling (7, 8) are one path now
    assert abs(d_arcs[0][0][3] + nxt[0][3] + d_final[nxt[0][0]] - total_b) < 1e-12
    assert abs(d_arcs[0][0][3] + nxt[1][3] + d_final[nxt[1][0]] - 2.0) < 1e-12
    m_arcs, m_final = G.minimize_encoded(d_arcs, d_final)
    assert len(m_arcs) == 3 and all(abs(w * 1024 - round(w * 1024)) < 1e-9 for row in m_arcs for (_d, _t, _o, w) in row)

    # (b) compiled graphs
    tm = fx.mono_tm
    plain = G.TrainingGraphCompiler(tm, fx.mono_tree, fx.mono_lex, determinize=False)
    det_c = G.TrainingGraphCompiler(tm, fx.mono_tree, fx.mono_lex, determinize=True)

    def paths(f, max_arcs):
        """(transition-ids without self-loops, words) → cost of every accepted path of at most ``max_arcs`` forward arcs
        (the silence model's inner states form cycles, so the language is infinite: compared up to a length)."""
        out = {}

        def walk(s, tids, words, cost):
            if len(tids) > max_arcs:
                return
            if np.isfinite(f.final[s]):
                key = (tuple(tids), tuple(words))
                c = cost + float(f.final[s])
                out[key] = min(out.get(key, np.inf), c)
            for a in f.arcs[f.arc_offsets[s]: f.arc_offsets[s + 1]]:
                if tm.is_self_loop[a["ilabel"]]:
                    continue
                walk(int(a["nextstate"]), tids + [int(a["ilabel"])], words + ([int(a["olabel"])] if a["olabel"] else []), cost + float(a["weight"]))

        walk(f.start, [], [], 0.0)
        return out

    for text, max_arcs in (("the", 13), ("this is", 18), ("i'm", 13)):
        a, b = plain.compile_fst(text), det_c.compile_fst(text)
        assert b.num_states < a.num_states
        for s in range(b.num_states):                                     # deterministic on the input side
            il = b.arcs["ilabel"][b.arc_offsets[s]: b.arc_offsets[s + 1]]
            assert len(set(il.tolist())) == len(il)
        pa, pb = paths(a, max_arcs), paths(b, max_arcs)
        assert len(pa) > 3 and set(pa) == set(pb)
        for k in pa:
            assert abs(pa[k] - pb[k]) <= (len(k[0]) + 1) / 2048 + 1e-6, (text, k)
    # (c) the same alignment through both
    scaled = tm.scaled_log_probs(1.0, 0.1)
    text = " ".join(fx.text.split()[:12])
    x = fx.mono_feats(fx.pcm[: 16000 * 5])
    ra = helpers.oracle_align_feats(tm, G.add_transition_probs(plain.compile_fst(text), scaled), x, fx.mono_am, beam=1e9, retry_beam=0.0)
    rb = helpers.oracle_align_feats(tm, G.add_transition_probs(det_c.compile_fst(text), scaled), x, fx.mono_am, beam=1e9, retry_beam=0.0)
    assert ra["status"] == 0 and rb["status"] == 0
    assert np.array_equal(ra["ali"], rb["ali"]) and np.array_equal(ra["words"], rb["words"])
    assert abs(ra["like"] - rb["like"]) < 10 * len(ra["ali"]) / 2048


def test_host_thread_budget_follows_the_cgroup_quota(tmp_path, monkeypatch):
    """hostcpu: the affinity mask alone over-counts inside a container (the GPU box: 256 cores in the mask, a 16-CPU quota)."""
    import math

    from montreal_forced_aligner_amd import hostcpu

    (tmp_path / "cpu.max").write_text("1600000 100000\n")
    assert hostcpu._cgroup_quota(str(tmp_path)) == 16.0
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert hostcpu._cgroup_quota(str(tmp_path)) == math.inf
    (tmp_path / "cpu.max").unlink()
    (tmp_path / "cpu").mkdir()
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("400000\n")
    (tmp_path / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert hostcpu._cgroup_quota(str(tmp_path)) == 4.0
    assert hostcpu._cgroup_quota(str(tmp_path / "nowhere")) == math.inf
    hostcpu.cpu_budget.cache_clear()
    monkeypatch.setenv("MFA_HOST_THREADS", "5")
    assert hostcpu.cpu_budget() == 5 and hostcpu.threads() == 5 and hostcpu.threads(0.5) == 2 and hostcpu.threads(1.0, cap=3) == 3
    hostcpu.cpu_budget.cache_clear()
    monkeypatch.delenv("MFA_HOST_THREADS")
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    whole = hostcpu.cpu_budget()
    assert 1 <= whole <= (len(__import__("os").sched_getaffinity(0)))
    hostcpu.cpu_budget.cache_clear()
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")                # four ranks on the node share its cores
    assert hostcpu.cpu_budget() == max(1, whole // 4)
    hostcpu.cpu_budget.cache_clear()


def test_negative_epsilon_cycle_detector():
    from montreal_forced_aligner_amd import kaldi_io as K
    from tests import helpers

    def fst(arcs_by_state):
        off = np.concatenate([[0], np.cumsum([len(a) for a in arcs_by_state])]).astype(np.int64)
        arcs = np.array([x for a in arcs_by_state for x in a], dtype=K.ARC_DTYPE)
        return K.Fst(0, off, arcs, np.zeros(len(arcs_by_state), dtype=np.float32))

    # 0 -eps/-0.5-> 1 -eps/0.1-> 0: a negative cycle; with +0.6 on the way back it is not
    assert helpers.has_negative_eps_cycle(fst([[(0, 0, -0.5, 1)], [(0, 0, 0.1, 0)]]))
    assert not helpers.has_negative_eps_cycle(fst([[(0, 0, -0.5, 1)], [(0, 0, 0.6, 0)]]))
    # the negative arc is emitting: no epsilon cycle at all
    assert not helpers.has_negative_eps_cycle(fst([[(3, 0, -0.5, 1)], [(0, 0, 0.1, 0)]]))
    # zero-weight epsilon cycles are fine (Kaldi replaces a token only when strictly cheaper)
    assert not helpers.has_negative_eps_cycle(fst([[(0, 0, 0.0, 1)], [(0, 0, 0.0, 0)]]))
