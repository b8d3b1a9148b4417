"""pytest plugin for running the REFERENCE's own acceptance suites (tests/test_flash_attn_ck.py, tests/test_flash_attn.py of Dao-AILab/flash-attention)
against our flash_attn_2_cuda module.  It is copied next to a git-ignored scratch copy of those files (tools/ref_suite/run.sh); nothing of the reference
is part of this repository.

The CK suite alone parametrises 412 614 cases (304 128 of them test_flash_attn_kvcache): far beyond one GPU session.  This plugin
  * takes a deterministic SAMPLE: a case runs iff crc32(nodeid) % stride(function) == 0, stride chosen so that every test function contributes
    about REF_SUITE_PER_FN cases (functions with fewer cases run in full); the same ids are picked on every box;
  * restricts generation to the functions named in REF_SUITE_FUNCS (comma list) so that parallel shards do not each pay for the 300k-case generator;
  * writes per-test-function pass / fail / skip counts and every failure's id + message to $REF_SUITE_OUT (json lines)."""
import json
import os
import zlib

import pytest

PER_FN = int(os.environ.get("REF_SUITE_PER_FN", "250"))
FUNCS = [f for f in os.environ.get("REF_SUITE_FUNCS", "").split(",") if f]
OUT = os.environ.get("REF_SUITE_OUT", "ref_suite_result.jsonl")
SEED = os.environ.get("REF_SUITE_SEED", "")   # a different string picks a different (equally deterministic) sample
DESELECT = os.environ.get("REF_SUITE_DESELECT") or None   # regex over node ids: cases that do not apply to a ROCm backend (stated in run.sh)
CAPABILITY = os.environ.get("REF_SUITE_FAKE_CAPABILITY")   # collection on a box without a GPU (counting cases only)
if CAPABILITY:
    import torch
    torch.cuda.get_device_capability = lambda *a, **k: (9, 0)


def pytest_pycollect_makeitem(collector, name, obj):
    if FUNCS and name.startswith("test_") and callable(obj) and name not in FUNCS:
        return []   # do not even generate this function's parameter grid
    return None


def pytest_collection_modifyitems(config, items):
    import re
    by_fn, not_applicable = {}, []
    rx = re.compile(DESELECT) if DESELECT else None
    for it in items:
        if rx is not None and rx.search(it.nodeid):
            not_applicable.append(it)
            continue
        by_fn.setdefault(it.originalname if hasattr(it, "originalname") else it.name, []).append(it)
    keep, drop, totals = [], [], {}
    for fn, its in by_fn.items():
        stride = max(1, len(its) // PER_FN)
        n = 0
        for it in its:
            if zlib.crc32((it.nodeid + SEED).encode()) % stride == 0:
                keep.append(it); n += 1
            else:
                drop.append(it)
        totals[fn] = {"parametrised": len(its), "sampled": n, "stride": stride}
    config._ref_totals = totals
    config._ref_not_applicable = len(not_applicable)
    drop += not_applicable
    if drop:
        config.hook.pytest_deselected(items=drop)
    items[:] = keep


_counts = {}
_fail = []


def pytest_runtest_logreport(report):
    fn = report.nodeid.split("::")[-1].split("[")[0]
    c = _counts.setdefault(fn, {"passed": 0, "failed": 0, "skipped": 0})
    if report.when == "call":
        if report.passed:
            c["passed"] += 1
        elif report.failed:
            c["failed"] += 1
            _fail.append({"id": report.nodeid, "msg": str(report.longrepr)[-1500:]})
    elif report.when == "setup":
        if report.skipped:
            c["skipped"] += 1
        elif report.failed:
            c["failed"] += 1
            _fail.append({"id": report.nodeid, "msg": "setup: " + str(report.longrepr)[-1500:]})
    if report.when == "call" and report.skipped:
        c["skipped"] += 1


def pytest_sessionfinish(session, exitstatus):
    if hasattr(session.config, "workerinput"):
        return   # xdist worker: the controller aggregates
    with open(OUT, "a") as f:
        f.write(json.dumps({"totals": getattr(session.config, "_ref_totals", {}), "counts": _counts, "failures": _fail[:200], "not_applicable": getattr(session.config, "_ref_not_applicable", 0), "deselect": DESELECT, "seed": SEED, "exit": int(exitstatus)}) + "\n")
