"""Host logic of the pipeline's job coalescing (sopro_amd/pipeline.py `_coalesce`), no GPU: which consecutive jobs merge into one
pass, and that every utterance keeps the sampler identity (nonce of its own job, index within it) it has when run alone."""
import types

from sopro_amd.pipeline import PipelinedSynthesizer


class _Model:
    def __init__(self):
        self.n = 100

    def next_nonce(self, seed=None):
        if seed is not None:
            return 7000 + int(seed)
        self.n += 1
        return self.n


def _stub():
    return types.SimpleNamespace(lanes=[types.SimpleNamespace(model=_Model())], _PER_UTT=PipelinedSynthesizer._PER_UTT)


def _job(n, **kw):
    j = dict(texts=[f"t{i}" for i in range(n)], refs=[object() for _ in range(n)], text_ids=[[i] for i in range(n)], max_frames=199, top_p=0.9,
             temperature=1.05, anti_loop=True)
    j.update(kw)
    return j


def test_consecutive_compatible_jobs_merge_in_groups_of_n():
    jobs = [_job(3), _job(2), _job(4), _job(1), _job(2)]
    passes = PipelinedSynthesizer._coalesce(_stub(), jobs, 2)
    assert [g for g, _m, _s in passes] == [[0, 1], [2, 3], [4]]
    g, merged, sizes = passes[0]
    assert sizes == [3, 2] and len(merged["refs"]) == 5 and merged["texts"] == ["t0", "t1", "t2", "t0", "t1"]
    assert merged["row_ids"] == [0, 1, 2, 0, 1]  # index within the utterance's own job
    assert merged["nonces"][:3] == [merged["nonces"][0]] * 3 and merged["nonces"][3:] == [merged["nonces"][3]] * 2
    assert merged["nonces"][0] != merged["nonces"][3]  # one nonce per job, as if each had run alone
    assert "seed" not in merged and merged["max_frames"] == 199
    assert passes[2][1] is jobs[4] and passes[2][2] is None  # a single job passes through untouched


def test_jobs_with_other_sampling_parameters_do_not_merge():
    jobs = [_job(2), _job(2, top_p=0.5), _job(2, top_p=0.5), _job(2, max_frames=99), _job(2)]
    groups = [g for g, _m, _s in PipelinedSynthesizer._coalesce(_stub(), jobs, 4)]
    assert groups == [[0], [1, 2], [3], [4]]


def test_a_seeded_job_keeps_its_seed_nonce_inside_a_merged_pass():
    jobs = [_job(2, seed=5), _job(3, seed=9)]
    (g, merged, sizes), = PipelinedSynthesizer._coalesce(_stub(), jobs, 2)
    assert g == [0, 1] and merged["nonces"] == [7005, 7005, 7009, 7009, 7009]


def test_jobs_with_different_per_utterance_keys_do_not_merge():
    """one job passing `texts`, its neighbour only `text_ids` (or omitting a list) must not end up in one pass whose
    merged lists have different lengths."""
    a, b, c = _job(2), _job(2), _job(2)
    b["texts"] = None
    del c["text_ids"]
    groups = [g for g, _m, _s in PipelinedSynthesizer._coalesce(_stub(), [a, b, b, c, c, a], 3)]
    assert groups == [[0], [1, 2], [3, 4], [5]]
    for g, merged, sizes in PipelinedSynthesizer._coalesce(_stub(), [b, b, c, c], 2):
        n = sum(sizes)
        assert all(len(merged[k]) == n for k in PipelinedSynthesizer._PER_UTT if merged.get(k) is not None)


def test_ramp_keeps_the_first_pass_single():
    """The first pass of a long run is one job (the pipeline's fill: the throughput partition gets work sooner); the rest merge."""
    jobs = [_job(2) for _ in range(7)]
    groups = [g for g, _m, _s in PipelinedSynthesizer._coalesce(_stub(), jobs, 2, ramp=True)]
    assert groups == [[0], [1, 2], [3, 4], [5, 6]]
    assert [g for g, _m, _s in PipelinedSynthesizer._coalesce(_stub(), jobs, 2)] == [[0, 1], [2, 3], [4, 5], [6]]


def test_pass_sizes_follow_the_queue_depth():
    """PipelinedSynthesizer.pass_sizes (round 5): an integer = that many jobs per pass; "auto" = 2 per pass for short queues, 4 from 4 queued jobs
    per lane; a list = explicit sizes; every job lands in exactly one pass."""
    from sopro_amd.pipeline import PipelinedSynthesizer

    p = PipelinedSynthesizer.__new__(PipelinedSynthesizer)
    p.lanes = [None] * 4
    assert p.pass_sizes(20, 2) == [2] * 10 and p.pass_sizes(7, 3) == [3, 3, 1] and p.pass_sizes(5, 1) == [1] * 5
    assert p.pass_sizes(20, "auto") == [4] * 5 and p.pass_sizes(12, "auto") == [2] * 6
    assert p.pass_sizes(32, "auto") == [4] * 8 and p.pass_sizes(35, "auto") == [4] * 8 + [3]
    assert p.pass_sizes(4, "auto") == [2, 2] and p.pass_sizes(1, "auto") == [1]
    assert p.pass_sizes(20, [2, 4, 4]) == [2, 4, 4, 4, 4, 2] and p.pass_sizes(5, [8]) == [5]
    for n in range(1, 70):
        assert sum(p.pass_sizes(n, "auto")) == n and min(p.pass_sizes(n, "auto")) >= 1
