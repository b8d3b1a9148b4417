"""Host logic of the dispatch, on a box without a GPU: fa_fwd_schedule_query / fa_bwd_dq_schedule_query run the same heuristics the launch path uses
(fa_api.cpp fwd_schedule_nw, pack_group, the feature / head-dim fallbacks, bwd_dq_schedule) and launch nothing.  The expectations are the measured
choices of profiles/r03_fwd_schedules.txt and the BASELINE.json configurations."""
import ctypes as C
import os

import pytest

from flash_attn_amd import _cabi


@pytest.fixture()
def lib(monkeypatch):
    for k in ("FA_FWD_NW", "FA_STRICT", "FA_BWD_DQ_NW", "FA_PACK_GQA"):
        monkeypatch.delenv(k, raising=False)
    L = _cabi.load()
    L.fa_knobs_reload()
    yield L
    for k in ("FA_FWD_NW", "FA_STRICT", "FA_BWD_DQ_NW", "FA_PACK_GQA"):
        os.environ.pop(k, None)
    L.fa_knobs_reload()


def fwd_params(B, Sq, Sk, H, Hk, D, causal=False, window=(-1, -1), bf16=True, **kw):
    a = _cabi.FaFwdParams()
    a.b, a.h, a.h_k, a.d = B, H, Hk, D
    a.seqlen_q, a.seqlen_k, a.total_q = Sq, Sk, B * Sq
    a.dtype = 1 if bf16 else 0
    a.is_causal, a.window_left, a.window_right = int(causal), window[0], window[1]
    a.softmax_scale = D ** -0.5
    a.k_row_stride = a.v_row_stride = Hk * D
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def q(lib, a, varlen=0):
    return lib.fa_fwd_schedule_query(C.byref(a), varlen)


def test_bf16_dtype_code():
    assert _cabi.FA_DTYPE_BF16 == 1 if hasattr(_cabi, "FA_DTYPE_BF16") else True


def test_baseline_configs_pick_the_measured_schedules(lib):
    assert q(lib, fwd_params(8, 2048, 2048, 16, 16, 64)) == 64                                   # config 2: D = 64, 32 key tiles, grid fills the chip
    assert q(lib, fwd_params(4, 4096, 4096, 32, 32, 128, causal=True)) == 64                     # config 3 (the headline)
    assert q(lib, fwd_params(2, 8192, 8192, 32, 8, 128, causal=True, window=(1024, 0))) == 34    # config 5: short windowed key range -> pipelined kernel
    a = fwd_params(16, 4096, 4096, 16, 16, 128, causal=True, cu_seqlens_q=1, cu_seqlens_k=1)     # config 4-i (pointers only count as flags here)
    a.total_q = 65536
    assert q(lib, a, varlen=1) == 64


def test_sweep_rows_and_grid_fill(lib):
    for S, want_nc, want_c in ((512, 64, 64), (1024, 64, 64), (2048, 64, 64), (4096, 64, 64), (16384, 64, 64)):   # (causal S = 512: the 64-rows kernel since round 5's peeled iterations)
        B = max(1, 16384 // S)
        assert q(lib, fwd_params(B, S, S, 16, 16, 128)) == want_nc, S
        assert q(lib, fwd_params(B, S, S, 16, 16, 128, causal=True)) == want_c, S
    assert q(lib, fwd_params(1, 1024, 1024, 4, 4, 128)) == 34          # 16 blocks of 256 rows cannot fill 256 CUs: 128-row blocks
    assert q(lib, fwd_params(1, 16384, 16384, 2, 2, 128)) == 64        # long key loops take it regardless
    assert q(lib, fwd_params(32, 512, 384, 16, 16, 128)) == 34         # 6 key tiles in all without a right bound: not measured on the 64-rows kernel, the old threshold stands
    assert q(lib, fwd_params(32, 512, 768, 16, 16, 128, causal=True)) == 64   # (6 visible tiles on average under the causal bound: measured at 4)
    a = fwd_params(32, 512, 512, 16, 16, 128, causal=True, cu_seqlens_q=1, cu_seqlens_k=1)   # a packed batch is sized by its longest sequence: the old threshold
    assert q(lib, a, varlen=1) == 34
    assert q(lib, fwd_params(64, 128, 4096, 16, 16, 128)) == 4         # short query chunks: lock-step
    assert q(lib, fwd_params(16, 1024, 1024, 16, 16, 64, causal=True)) == 34   # D = 64 under a causal mask needs 16 tiles on average
    assert q(lib, fwd_params(8, 2048, 2048, 16, 16, 64, causal=True)) == 64


def test_feature_and_head_dim_fallbacks(lib):
    big = dict(B=4, Sq=4096, Sk=4096, H=32, Hk=32, D=128, causal=True)
    assert q(lib, fwd_params(**big, softcap=30.0)) == 64                # round 5: the softcap variant of the 64-rows-per-wave kernel
    assert q(lib, fwd_params(**big, p_dropout=0.1)) == 64               # ... and its dropout variant
    assert q(lib, fwd_params(**big, p_dropout=0.1, randval=1)) == 8     # the random-byte output (return_softmax) and feature products: 8-wave lock-step on the same 256-row blocks
    assert q(lib, fwd_params(**big, alibi_slopes=1)) == 64               # causal ALiBi: the variant of the 64-rows-per-wave kernel
    assert q(lib, fwd_params(**big, alibi_slopes=1, bf16=False)) == 64
    assert q(lib, fwd_params(4, 4096, 4096, 32, 32, 128, alibi_slopes=1)) == 8   # not causal: |i - j| is not linear in j
    assert q(lib, fwd_params(**big, alibi_slopes=1, softcap=30.0)) == 8
    assert q(lib, fwd_params(**big, p_dropout=0.1, softcap=30.0)) == 8
    assert q(lib, fwd_params(**big, block_table=1, page_block_size=256)) == 64   # round 5: a paged cache runs on the 64-rows-per-wave kernel (a descriptor per tile)
    assert q(lib, fwd_params(**big, block_table=1, page_block_size=256, softcap=30.0)) == 8   # ... plain attention only
    for d in (32, 96, 192, 256, 72, 160):                               # trimmed / bounded / 256: the 4-wave lock-step kernel
        assert q(lib, fwd_params(4, 4096, 4096, 32, 32, d, causal=True)) == 4, d
    assert q(lib, fwd_params(2, 1024, 1024, 8, 8, 128, softcap=30.0)) == 4   # a pipelined choice with a feature: lock-step with the same wave count
    assert q(lib, fwd_params(2, 16, 2048, 32, 8, 128)) == 4             # grouped heads of a short chunk are packed: 4-wave lock-step
    assert q(lib, fwd_params(4, 4096, 4096, 32, 32, 200)) == 4         # between the built sizes: column-bounded, the 256 kernel
    assert q(lib, fwd_params(4, 4096, 4096, 32, 32, 100)) < 0          # not a multiple of 8
    assert q(lib, fwd_params(4, 4096, 4096, 32, 32, 264)) < 0          # beyond 256


def test_knobs_override_and_strict(lib, monkeypatch):
    a = fwd_params(4, 4096, 4096, 32, 32, 128, causal=True)
    monkeypatch.setenv("FA_STRICT", "1"); lib.fa_knobs_reload()
    assert q(lib, a) == 34                                              # fp32 scaling of every score: pipelined kernel (32 tiles on average)
    assert q(lib, fwd_params(1, 16384, 16384, 16, 16, 128)) == 38
    monkeypatch.delenv("FA_STRICT"); monkeypatch.setenv("FA_FWD_NW", "38"); lib.fa_knobs_reload()
    assert q(lib, a) == 38
    monkeypatch.setenv("FA_FWD_NW", "64"); lib.fa_knobs_reload()
    assert q(lib, fwd_params(2, 300, 300, 4, 4, 128)) == 64
    huge = fwd_params(1, 131072, 131072, 128, 128, 128)                 # K/V rows 32 KB apart: the key range spans > 4 GiB
    assert q(lib, huge) == 64                                           # forced: the knob stands, the launcher then refuses with its own message (-3)
    monkeypatch.delenv("FA_FWD_NW"); lib.fa_knobs_reload()
    assert q(lib, huge) == 38                                           # the heuristic itself falls back to the pipelined kernel


def test_backward_dq_schedule(lib, monkeypatch):
    def bp(B, S, H, D, **kw):
        a = _cabi.FaBwdParams()
        a.b, a.h, a.h_k, a.d = B, H, H, D
        a.seqlen_q = a.seqlen_k = S
        a.total_q = a.total_k = B * S
        a.dtype = 1
        a.softmax_scale = D ** -0.5
        a.window_left = a.window_right = -1
        for k, v in kw.items():
            setattr(a, k, v)
        return a
    dq = lambda a: lib.fa_bwd_dq_schedule_query(C.byref(a))
    assert dq(bp(4, 4096, 32, 128)) == 64
    assert dq(bp(4, 1024, 32, 128, is_causal=1)) == 4
    # (late round 6) without a right bound from 768 keys, once the 256-row blocks fill the chip (>= 256 of them), plain attention only
    assert dq(bp(4, 1024, 32, 128)) == 64 and dq(bp(8, 768, 16, 128)) == 64 and dq(bp(8, 640, 16, 128)) == 4 and dq(bp(1, 1024, 16, 128)) == 4
    assert dq(bp(4, 1024, 32, 128, softcap=20.0)) == 4 and dq(bp(4, 1024, 32, 128, window_left=100)) == 4
    assert dq(bp(4, 4096, 32, 64)) == 64                                 # head dim 64 (round 4): from ~2k visible keys per row on average
    assert dq(bp(8, 2048, 16, 64)) == 64                                 # config 2's shape
    assert dq(bp(4, 4096, 32, 64, is_causal=1)) == 64                    # (a tie there)
    assert dq(bp(8, 2048, 32, 64, is_causal=1)) == 4
    assert dq(bp(2, 8192, 32, 64, is_causal=1)) == 64
    assert dq(bp(16, 1024, 32, 64)) == 4 and dq(bp(16, 1536, 32, 64)) == 64 and dq(bp(16, 1536, 32, 64, is_causal=1)) == 4   # (late round 6: without a right bound from 1536 keys)
    assert dq(bp(4, 4096, 32, 128, softcap=20.0)) == 64                   # round 5: softcap / dropout variants of the 64-rows-per-wave dQ kernel at head dim 128
    assert dq(bp(4, 4096, 32, 128, p_dropout=0.1)) == 64
    assert dq(bp(4, 4096, 32, 64, softcap=20.0)) == 4                     # ... not at head dim 64 (the 4-wave feature kernel measured ahead), not for products of features
    assert dq(bp(4, 4096, 32, 128, softcap=20.0, p_dropout=0.1)) == 4
    assert dq(bp(4, 4096, 32, 128, alibi_slopes=1, is_causal=1)) == 64    # round 5: ALiBi under a causal bound runs on the 64-rows-per-wave kernel too
    assert dq(bp(4, 4096, 32, 128, alibi_slopes=1)) == 4                  # ... not without it (|key - row| is not linear in the key)
    monkeypatch.setenv("FA_BWD_DQ_NW", "64"); lib.fa_knobs_reload()
    assert dq(bp(4, 4096, 32, 96)) == 4                                 # trimmed head dims only have the 4-wave kernel, whatever the knob says
    assert dq(bp(4, 512, 8, 128)) == 64


def test_fused_backward_workspace_and_conditions(lib, monkeypatch):
    """FA_BWD_MODE=3 (opt-in, fa_api.cpp bwd_fused_ds_bytes): the workspace is the dS matrix of every (batch, head) in 2-KB sub-tiles plus the sync area
    (fa_kernel_params.h fz_sync_words: 1344 words of control blocks, one 128-byte line per arrival counter, eight queues); calls it does not cover ask for
    nothing, and so does every call without the knob."""
    def bp(B, Sq, Sk, H, Hk, D, **kw):
        a = _cabi.FaBwdParams()
        a.b, a.h, a.h_k, a.d = B, H, Hk, D
        a.seqlen_q, a.seqlen_k, a.total_q, a.total_k = Sq, Sk, B * Sq, B * Sk
        a.dtype = 1
        a.softmax_scale = D ** -0.5
        a.window_left = a.window_right = -1
        for k, v in kw.items():
            setattr(a, k, v)
        return a
    ws = lambda a: lib.fa_bwd_workspace_bytes(C.byref(a))
    assert ws(bp(4, 4096, 4096, 32, 32, 128)) == 0                        # default: the scratch-free 7-contraction pair
    monkeypatch.setenv("FA_BWD_MODE", "3"); lib.fa_knobs_reload()
    try:
        def expect(B, Sq, Sk, H, causal=False):
            # rows packed in 64-key pairs (round 6, csrc/fa_device.h ds_row_start): row block i holds the pairs up to its last visible key sub-tile
            np64, tiles = (Sk + 63) // 64, 0
            for i in range((Sq + 31) // 32):
                last32 = (32 * i + 31 + (Sk - Sq)) // 32 if causal else 10 ** 9
                tiles += 2 * min(np64, last32 // 2 + 1)
            ds = (B * H * tiles * 2048 + 255) & ~255
            items = B * H * ((Sq + 255) // 256)
            return ds + (1344 + items * 32 + 8 * items) * 4
        assert ws(bp(4, 4096, 4096, 32, 32, 128)) == expect(4, 4096, 4096, 32)    # config 3 without a mask: 4.3 GB
        assert ws(bp(4, 4096, 4096, 32, 32, 128, is_causal=1)) == expect(4, 4096, 4096, 32, True) < 0.52 * expect(4, 4096, 4096, 32)   # the causal triangle: half
        assert ws(bp(1, 256, 256, 2, 2, 128)) == expect(1, 256, 256, 2)
        assert ws(bp(2, 1000, 1024, 32, 8, 64, is_causal=1)) == expect(2, 1000, 1024, 32, True)
        assert ws(bp(1, 16384, 16384, 32, 32, 128)) == 0                    # 17 GB > FA_BWD_DS_CAP_MB (8192)
        assert ws(bp(4, 4096, 4096, 32, 32, 256)) == 0                      # head dim
        assert ws(bp(4, 4096, 4096, 32, 32, 128, window_left=1000)) == 0    # left window
        assert ws(bp(4, 4096, 4096, 32, 32, 128, softcap=30.0)) == 0
        assert ws(bp(4, 4096, 4096, 32, 32, 128, p_dropout=0.1)) == 0
        assert ws(bp(4, 4096, 1024, 32, 32, 128, is_causal=1)) == 0          # sk < sq
        monkeypatch.setenv("FA_BWD_DS_CAP_MB", "1024"); lib.fa_knobs_reload()
        # (round 6, late: a batch that does not fit the cap is cut into chunks of whole batch entries, one launch each on the same workspace -- here one entry = 1 GiB)
        assert ws(bp(4, 4096, 4096, 32, 32, 128)) == expect(1, 4096, 4096, 32) and ws(bp(16, 1024, 1024, 32, 32, 128)) == expect(16, 1024, 1024, 32)
        monkeypatch.setenv("FA_BWD_DS_CAP_MB", "512"); lib.fa_knobs_reload()
        assert ws(bp(4, 4096, 4096, 32, 32, 128)) == 0                      # not even one batch entry fits
    finally:
        monkeypatch.delenv("FA_BWD_MODE", raising=False); monkeypatch.delenv("FA_BWD_DS_CAP_MB", raising=False); lib.fa_knobs_reload()


def _bwd_params(B, Sq, Sk, H, Hk, D, **kw):
    a = _cabi.FaBwdParams()
    a.b, a.h, a.h_k, a.d = B, H, Hk, D
    a.seqlen_q, a.seqlen_k, a.total_q, a.total_k = Sq, Sk, B * Sq, B * Sk
    a.dtype = 1
    a.softmax_scale = D ** -0.5
    a.window_left = a.window_right = -1
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _plan(lib, a):
    out = (C.c_int32 * 8)()
    assert lib.fa_bwd_plan_query(C.byref(a), out, 8) == 8
    return list(out)


def test_backward_plan_table(lib, monkeypatch):
    """Round 6 (fa_api.cpp bwd_fused_by_table): the fused 5-contraction launch is the default at head dim 128 with Sq = Sk and at least 32 (batch, kv head) units, under a
    causal mask from 512 to 4096 rows (from 256 rows on large grids, there also without a mask), while its dS workspace fits 1.25 GiB (whole, or up to 2048 rows in chunks of batch entries) -- where it was measured ahead -- and nowhere
    else; knobs that pin a kernel of the recomputing pair keep the pair."""
    for v in ("FA_BWD_MODE", "FA_BWD_DQ_NW", "FA_BWD_DKDV", "FA_STRICT"):
        monkeypatch.delenv(v, raising=False)
    lib.fa_knobs_reload()
    plan = lambda *a, **kw: _plan(lib, _bwd_params(*a, **kw))[0]
    assert plan(16, 1024, 1024, 16, 16, 128, is_causal=1) == 3            # the sweep's rows: 0.5 GiB / 1 GiB of dS
    assert plan(8, 2048, 2048, 16, 16, 128, is_causal=1) == 3
    assert plan(8, 2048, 2048, 32, 32, 128, is_causal=1) == 3             # 1.03 GiB of packed rows (2 GiB as a square): inside the 1.25 GiB bound, +15 %
    assert plan(32, 512, 512, 16, 16, 128, is_causal=1) == 3              # +8 % at S = 512
    assert plan(64, 256, 256, 16, 16, 128, is_causal=1) == 3 and plan(32, 256, 256, 16, 16, 128, is_causal=1) == 0   # S = 256: +4 % on 1024 units, a tie on 512 (units x rows >= 196608)
    assert plan(32, 384, 384, 16, 16, 128, is_causal=1) == 3 and plan(16, 384, 384, 16, 16, 128, is_causal=1) == 0   # S = 384: +7 % on 512 units, a tie on 256
    assert plan(256, 128, 128, 32, 32, 128, is_causal=1) == 0             # S = 128: -5 %
    assert plan(64, 384, 384, 32, 32, 128) == 3 and plan(32, 384, 384, 16, 16, 128) == 0   # no mask below 512 rows: from 2048 units at S = 384 (+7 %)
    assert plan(4, 4096, 4096, 32, 32, 128, is_causal=1) == 0             # config 3: 4 GiB of dS -- over the bound (and a tie): the scratch-free pair
    assert plan(1, 4096, 4096, 32, 32, 128, is_causal=1) == 3             # the same rows on 32 heads fit: +9 ... +13 %
    assert plan(2, 3072, 3072, 32, 32, 128, is_causal=1) == 3 and plan(4, 3072, 3072, 32, 32, 128, is_causal=1) == 3   # (0.6 / 1.2 GiB of packed rows: +9 % / +7 %)
    assert plan(8, 3072, 3072, 32, 32, 128, is_causal=1) == 0             # 2.4 GiB: above 2048 rows the whole batch has to fit (the pair gains more from the larger grid)
    # up to 2048 rows a batch over the bound is cut into chunks of whole batch entries (>= 32 units each), one launch per chunk on the same workspace
    full = lambda *a, **kw: _plan(lib, _bwd_params(*a, **kw))
    assert full(32, 1024, 1024, 32, 32, 128, is_causal=1)[:3] == [3, 1, 32]   # 1.09 GiB of packed rows: one launch (+25 %)
    assert full(64, 1024, 1024, 32, 32, 128, is_causal=1)[:3] == [3, 2, 32]   # 2.2 GiB -> two chunks of 32 batch entries
    assert full(16, 2048, 2048, 32, 32, 128, is_causal=1)[:3] == [3, 2, 8]    # 2.06 GiB (+12 %)
    assert full(65, 1024, 1024, 32, 32, 128, is_causal=1)[:3] == [3, 2, 33] and full(65, 1024, 1024, 32, 32, 128, is_causal=1)[7] <= 1280
    assert full(16, 1024, 1024, 16, 16, 128, is_causal=1)[:3] == [3, 1, 16]
    assert plan(64, 2048, 2048, 32, 1, 128, is_causal=1) == 0              # MQA: nine batch entries fit = nine (batch, kv head) units per chunk: the pair
    assert plan(4, 1536, 1536, 32, 32, 128) == 0                          # without a mask from 512 rows: the pair (its dQ half on 64 rows per wave from 768 keys is level or ahead, at no workspace)
    assert plan(2, 8192, 8192, 16, 16, 128, is_causal=1) == 0
    assert plan(16, 1024, 1024, 16, 16, 128) == 0 and plan(32, 512, 512, 16, 16, 128) == 0
    assert plan(8, 2048, 2048, 16, 16, 128) == 0
    assert plan(2, 2048, 2048, 32, 8, 128, is_causal=1) == 0              # 16 (batch, kv head) units: measured behind
    assert plan(4, 1024, 1024, 32, 8, 128, is_causal=1) == 3              # 32 units, four query heads each: +8 %
    assert plan(16, 1024, 1024, 32, 32, 64, is_causal=1) == 0             # ... and at head dim 64
    assert plan(16, 1024, 1024, 16, 16, 128, is_causal=1, window_left=256) == 0
    assert plan(16, 1024, 1024, 16, 16, 128, is_causal=1, softcap=30.0) == 0
    assert plan(16, 1024, 2048, 16, 16, 128, is_causal=1) == 0            # sq != sk: not measured
    a = _bwd_params(16, 1024, 1024, 16, 16, 128, is_causal=1)
    assert 16 * 16 * 32 * 32 * 2048 > lib.fa_bwd_workspace_bytes(C.byref(a)) >= 16 * 16 * (32 * 33 // 2) * 2048   # the binders size the workspace from this: the causal triangle, not the square
    for knob, val in (("FA_BWD_MODE", "-1"), ("FA_BWD_MODE", "1"), ("FA_BWD_DQ_NW", "4"), ("FA_BWD_DKDV", "8"), ("FA_STRICT", "1")):
        monkeypatch.setenv(knob, val); lib.fa_knobs_reload()
        try:
            assert plan(16, 1024, 1024, 16, 16, 128, is_causal=1) == 0, (knob, val)
        finally:
            monkeypatch.delenv(knob); lib.fa_knobs_reload()


def test_chunked_five_contraction_plan(lib, monkeypatch):
    """FA_BWD_MODE=5 (fa_api.cpp bwd_c5_plan): chunks of whole XCD rounds under FA_BWD_C5_CAP_MB, two slots; rows of a slot packed in 64-key pairs
    (csrc/fa_device.h ds_row_start): a head's sub-tile count must equal the sum over its 32-row blocks of the pairs each can see."""
    monkeypatch.setenv("FA_BWD_MODE", "5"); lib.fa_knobs_reload()
    try:
        def pairs_seen(Sq, Sk, wr):   # brute force: per 32-row block, the 64-key pairs up to the last visible key sub-tile
            np64, tot = (Sk + 63) // 64, 0
            for i in range((Sq + 31) // 32):
                last32 = (Sk + 31) // 32 - 1 if wr < 0 else min((Sk + 31) // 32 - 1 + 10 ** 9, (32 * i + 31 + (Sk - Sq) + wr) // 32)
                tot += min(np64, last32 // 2 + 1)
            return 2 * tot
        for (B, Sq, Sk, H, Hk, D, causal, wr) in ((4, 4096, 4096, 32, 32, 128, 1, -1), (2, 1000, 1024, 32, 8, 64, 1, -1), (1, 300, 333, 2, 2, 128, 0, -1), (1, 640, 900, 2, 2, 128, 0, 100),
                                                  (3, 1536, 1536, 8, 8, 128, 1, -1), (1, 256, 256, 1, 1, 64, 0, -1), (1, 777, 1000, 3, 1, 128, 0, 37)):
            p = _plan(lib, _bwd_params(B, Sq, Sk, H, Hk, D, is_causal=causal, window_right=wr))
            assert p[0] == 5, p
            assert p[3] == pairs_seen(Sq, Sk, 0 if causal else wr), (B, Sq, Sk, causal, wr, p)
            assert p[4] == (Sk + 63) // 64
            rounds = (B * Hk + 7) // 8
            assert p[1] * p[2] >= rounds and (p[1] - 1) * p[2] < rounds                     # the chunks cover the rounds, none is empty
            slot = 8 * p[2] * (H // Hk) * p[3] * 2048
            assert 2 * slot <= 1024 << 20 and p[7] == (((slot + 255) & ~255) >> 20)            # within the default bound
            a = _bwd_params(B, Sq, Sk, H, Hk, D, is_causal=causal, window_right=wr)
            assert lib.fa_bwd_workspace_bytes(C.byref(a)) == 2 * ((slot + 255) & ~255)
        p = _plan(lib, _bwd_params(4, 4096, 4096, 32, 32, 128, is_causal=1))
        assert p[1] >= 4                                                                      # config 3: 2.1 GB of (causal) dS through a 1 GiB workspace
        # what it does not cover asks for nothing and keeps the pair
        for kw in (dict(window_left=100), dict(softcap=30.0), dict(p_dropout=0.1), dict(alibi_slopes=1)):
            assert _plan(lib, _bwd_params(4, 4096, 4096, 32, 32, 128, **kw))[0] == 0
        assert _plan(lib, _bwd_params(4, 4096, 1024, 32, 32, 128, is_causal=1))[0] == 0       # sk < sq
        assert _plan(lib, _bwd_params(1, 32768, 32768, 8, 1, 128))[0] == 0                    # one round of units (8 x 8 heads x 2 GiB) does not fit a slot
        monkeypatch.setenv("FA_BWD_C5_CAP_MB", "128"); lib.fa_knobs_reload()   # (a round of this shape: 8 units x 4 heads x 1.1 MB of packed rows = 36 MB)
        assert _plan(lib, _bwd_params(2, 1024, 1024, 32, 8, 128, is_causal=1))[1] >= 2
    finally:
        monkeypatch.delenv("FA_BWD_MODE", raising=False); monkeypatch.delenv("FA_BWD_C5_CAP_MB", raising=False); lib.fa_knobs_reload()


def test_gqa_group_split_plan(lib, monkeypatch):
    """fa_api.cpp bwd_gsplit_plan (late round 6): the dK/dV kernels split a GQA group into 2 / 4 / 8 virtual kv heads while their (batch, kv head, key block) grid has
    fewer than 1024 items under a right bound alone, fewer than 256 with a left window or no mask; the workspace holds the partial dK and dV in the input dtype."""
    for v in ("FA_BWD_MODE", "FA_BWD_GSPLIT"):
        monkeypatch.delenv(v, raising=False)
    lib.fa_knobs_reload()
    full = lambda *a, **kw: _plan(lib, _bwd_params(*a, **kw))
    ws = lambda *a, **kw: lib.fa_bwd_workspace_bytes(C.byref(_bwd_params(*a, **kw)))
    assert full(2, 1024, 1024, 32, 2, 128, is_causal=1)[:4] == [0, 0, 0, 8]      # 16 items: eight virtual heads per group of 16
    assert ws(2, 1024, 1024, 32, 2, 128, is_causal=1) == 2 * 2 * 1024 * 2 * 8 * 128 * 2
    assert full(4, 4096, 4096, 32, 8, 128, is_causal=1)[3] == 2                    # 512 uneven items: two
    assert full(8, 4096, 4096, 32, 8, 128, is_causal=1)[3] == 0 and ws(8, 4096, 4096, 32, 8, 128, is_causal=1) == 0
    assert full(2, 8192, 8192, 32, 8, 128, is_causal=1, window_left=1024)[3] == 0   # config 5: 512 uniform items (a split in two costs 6 %)
    assert full(1, 2048, 2048, 32, 2, 128, is_causal=1, window_left=512)[3] == 8    # 16 items: split, window or not
    assert full(2, 4096, 4096, 32, 8, 128)[3] == 0 and full(4, 2048, 2048, 32, 4, 128)[3] == 2   # no mask: 256 items unsplit, 128 split in two
    assert full(2, 1024, 1024, 6, 2, 128, is_causal=1)[3] == 0                     # a group of three: no power of two divides it
    assert full(2, 1024, 1024, 8, 8, 128, is_causal=1)[3] == 0                     # no group
    a = _bwd_params(2, 1024, 1024, 32, 2, 128, is_causal=1); a.cu_seqlens_q = a.cu_seqlens_k = 1
    assert _plan(lib, a)[3] == 0                                                   # packed batches: not split
    monkeypatch.setenv("FA_BWD_GSPLIT", "0"); lib.fa_knobs_reload()
    assert full(2, 1024, 1024, 32, 2, 128, is_causal=1)[3] == 0 and ws(2, 1024, 1024, 32, 2, 128, is_causal=1) == 0
    monkeypatch.setenv("FA_BWD_GSPLIT", "4"); lib.fa_knobs_reload()
    assert full(8, 4096, 4096, 32, 8, 128, is_causal=1)[3] == 4                    # forced (tests): up to the knob, never past the group
    monkeypatch.delenv("FA_BWD_GSPLIT"); lib.fa_knobs_reload()
