"""Build check on the compiler's resource report (Makefile: -Rpass-analysis=kernel-resource-usage ->
approximate-spmv-topk_amd/kernel_resources.txt): the streaming kernels must not spill, and the ones launched as two
576-thread workgroups per CU must stay within 80 registers (at 81+ the dispatcher places one workgroup
per CU and every query takes twice as long -- a regression no functional test sees)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "approximate-spmv-topk_amd", "kernel_resources.txt")


def _report():
    if not os.path.exists(REPORT):
        pytest.skip("no resource report (the library was not built by this Makefile)")
    kernels, cur = {}, None
    for ln in open(REPORT):
        m = re.match(r"\s*Function Name: (\S+)", ln)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.match(r"\s*(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill): (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return kernels


def test_streaming_kernels_do_not_spill_and_fit_two_workgroups_per_cu():
    k = _report()
    stream = {n: v for n, v in k.items() if re.search(r"tkspmv1[23](stream|batch)_kernel", n)}
    multi = {n: v for n, v in k.items() if "tkspmv12multi_kernel" in n}
    assert len(stream) >= 30 and len(multi) == 12
    for n, v in {**stream, **multi}.items():
        tracing = re.search(r"stream_kernelILi4ELb0ELi1024ELi7ELi3ELb1E", n) or re.search(r"batch_kernelILi4ELi1024ELi[07]ELb1E", n)
        if "kernelILi8E" not in n and not tracing:  # (the opt-in 8-entries-per-lane variants sit at the register limit; so do the tracing instantiations (TKSPMV_TRACE / TKSPMV_STATS runs only): the 12-bit layout's stream kernel, the batch kernels since round 3's checked thresholds)
            assert v["VGPRs Spill"] == 0, n
        assert v["AGPRs"] == 0, n
    for n, v in stream.items():
        dbg = bool(re.search(r"stream_kernelILi4ELb0ELi1024ELi0ELi3ELb1E", n))  # the tracing instantiation may keep a few stamps in scratch
        scores = "stream_kernelILi4ELb1E" in n or "stream_kernelILi8ELb1E" in n  # SpMV-only variants: one workgroup per CU is fine
        c8 = "kernelILi8E" in n
        dbg = dbg or bool(re.search(r"stream_kernelILi4ELb0ELi1024ELi7ELi3ELb1E", n)) or bool(re.search(r"batch_kernelILi4ELi1024ELi[07]ELb1E", n))
        if not dbg and not c8:
            assert v["ScratchSize [bytes/lane]"] == 0, (n, v)
        if not scores:
            assert v["VGPRs"] <= 80, (n, v)
    # the headline kernels by name
    head = [n for n in stream if "12batch_kernelILi4ELi1024ELi7ELb0ELb0E" in n or "13stream_kernelILi4ELb0ELi1024ELi7ELi3ELb0E" in n
            or "12batch_kernelILi4ELi1024ELi7ELb0ELb1E" in n]
    assert len(head) == 3
    for n, v in multi.items():
        q8 = "multi_kernelILi8E" in n
        assert v["VGPRs"] <= (128 if q8 else 80), (n, v)  # 8 queries per pass run 8-wave workgroups
