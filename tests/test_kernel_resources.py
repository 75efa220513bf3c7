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
        fused_tail = re.search(r"stream_kernelILi4ELb0ELi1024ELi[07]ELi3ELb[01]E", n)  # (a word of the selection tail; the loop is checked on the ISA below)
        if "kernelILi8E" not in n and not tracing and "batch_kernel" not in n and not fused_tail:  # (batch kernels: the ISA test below; the opt-in 8-entries-per-lane variants sit at the register limit; so do the tracing instantiations (TKSPMV_TRACE / TKSPMV_STATS runs only): the 12-bit layout's stream kernel, the batch kernels since round 3's checked thresholds)
            assert v["VGPRs Spill"] == 0, n
        assert v["AGPRs"] == 0, n
    for n, v in stream.items():
        dbg = bool(re.search(r"stream_kernelILi4ELb0ELi1024ELi0ELi3ELb1E", n))  # the tracing instantiation may keep a few stamps in scratch
        scores = "stream_kernelILi4ELb1E" in n or "stream_kernelILi8ELb1E" in n  # SpMV-only variants: one workgroup per CU is fine
        c8 = "kernelILi8E" in n
        dbg = dbg or bool(re.search(r"stream_kernelILi4ELb0ELi1024ELi7ELi3ELb1E", n)) or bool(re.search(r"batch_kernelILi4ELi1024ELi[07]ELb1E", n))
        # (the batch kernels call their selections and their repair phase as functions since round 4: the scratch size is those
        #  functions' stack; that the streaming loop itself touches no scratch is checked on the ISA below)
        if not dbg and not c8 and "batch_kernel" not in n:
            # (the single-query kernels keep a word or two of their selection tail in scratch; their loops are checked on the ISA too)
            assert v["ScratchSize [bytes/lane]"] <= (16 if "stream_kernelILi4ELb0ELi1024" in n else 0), (n, v)
        if not scores and not dbg:  # (the tracing instantiations of the single-query kernel may run one workgroup per CU)
            assert v["VGPRs"] <= 80, (n, v)
    # the headline kernels by name
    head = [n for n in stream if "12batch_kernelILi4ELi1024ELi7ELb0ELb0E" in n or "13stream_kernelILi4ELb0ELi1024ELi7ELi3ELb0E" in n
            or "12batch_kernelILi4ELi1024ELi7ELb0ELb1E" in n]
    assert len(head) == 3  # (the batch kernel of local thresholds, the exact one, the single-query stream kernel)
    for n, v in multi.items():
        q8 = "multi_kernelILi8E" in n
        assert v["VGPRs"] <= (128 if q8 else 80), (n, v)  # 8 queries per pass run 8-wave workgroups


def test_batch_kernels_stream_without_touching_scratch(tmp_path):
    """The batch kernel sits AT its register limit (80: two 576-thread workgroups per CU). Round 4 found out what one value too
    many costs: three reloads from scratch memory per packet and twice the time per query, with every functional test green.
    This compiles the instantiations on their own (seconds) and reads the ISA: no basic block that requests a packet
    (non-temporal load) or runs the fp32 segmented scan (v_add_f32_dpp) may contain a scratch instruction."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    tu = tmp_path / "tu.hip"
    tu.write_text("""#include <hip/hip_runtime.h>
#include <cstdint>
#include "kernels/common.hpp"
#include "kernels/select.hpp"
#include "kernels/packet_math.hpp"
#include "kernels/stream_kernel.hpp"
#include "kernels/local.hpp"
#include "kernels/batch_kernel.hpp"
namespace tkspmv {
#define INST(QM) template __global__ void batch_kernel<4, 1024, QM, false, false>(const BatchArgs); template __global__ void batch_kernel<4, 1024, QM, false, true>(const BatchArgs);
INST(0) INST(1) INST(2) INST(3) INST(4) INST(5) INST(6) INST(7) INST(8)
template __global__ void stream_kernel<4, false, 1024, 7, 3, false>(const StreamParams, const SelectParams);
template __global__ void stream_kernel<4, false, 1024, 0, 3, false>(const StreamParams, const SelectParams);
}
""")
    asm = tmp_path / "tu.s"
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S",
                           "-I" + os.path.join(ROOT, "approximate-spmv-topk_amd", "csrc"), "-o", str(asm), str(tu)],
                          stderr=subprocess.DEVNULL)
    lines = asm.read_text().split("\n")
    starts = [i for i, ln in enumerate(lines) if (ln.startswith("_ZN6tkspmv12batch_kernel") or ln.startswith("_ZN6tkspmv13stream_kernel")) and "@" in ln]
    assert len(starts) == 20
    for start in starts:
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        blocks, cur = [], None
        for ln in lines[start:end]:
            if re.match(r"^\.LBB\d+_\d+:", ln):
                cur = {"name": ln.split(":")[0], "scratch": 0, "hot": False, "dpp": False, "max3": False}
                blocks.append(cur)
            elif cur is not None:
                if "scratch_" in ln:
                    cur["scratch"] += 1
                if ("global_load_dword" in ln or "buffer_load_dword" in ln) and " nt" in ln:  # (a packet request -- flat or, since round 5, buffer loads)
                    cur["hot"] = True
                # (the fp32 scan: DPP adds WITH the trigger's v_max3 behind them -- the server wave's sums of x use DPP adds too, round 5)
                cur["dpp"] = cur["dpp"] or "v_add_f32_dpp" in ln
                cur["max3"] = cur["max3"] or "v_max3_f32" in ln
        for b in blocks:
            b["hot"] = b["hot"] or (b["dpp"] and b["max3"])
        hot = [b for b in blocks if b["hot"]]
        assert len(hot) >= 3, "the streaming loop was not found in the ISA of " + lines[start].split(":")[0]
        assert all(b["scratch"] == 0 for b in hot), (lines[start].split(":")[0], [b for b in hot if b["scratch"]])


def test_no_chain_of_loads_waited_for_one_by_one(tmp_path):
    """Round 4 found the server wave staging x with sixteen loads each behind its own `s_waitcnt vmcnt(0)` -- sixteen trips through
    memory in a row, 6-16 us per query, invisible to every functional test (the compiler had put each 'in range ? load : 0' into a
    branch of its own). This compiles the headline kernels on their own and looks for runs of (wait for everything, ONE load) in
    the ISA: none of 6 or more may exist in the batch kernels (x staged with buffer loads since round 5: the resource's bounds check
    replaces clamped addresses, narrow x included), the single-query kernel and the multi-query kernels."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    tu = tmp_path / "tu.hip"
    tu.write_text("""#include <hip/hip_runtime.h>
#include <cstdint>
#include "kernels/common.hpp"
#include "kernels/select.hpp"
#include "kernels/packet_math.hpp"
#include "kernels/stream_kernel.hpp"
#include "kernels/local.hpp"
#include "kernels/batch_kernel.hpp"
#include "kernels/multi_kernel.hpp"
namespace tkspmv {
template __global__ void batch_kernel<4, 1024, 7, false, true>(const BatchArgs);
template __global__ void batch_kernel<4, 1024, 7, false, false>(const BatchArgs);
template __global__ void batch_kernel<4, 1024, 3, false, true>(const BatchArgs);
template __global__ void single_kernel<7>(const StreamParams, const SelectParams, const LocalParams);
template __global__ void multi_kernel<8, 0>(const StreamParams, const SelectParams, const MultiParams);
template __global__ void multi_kernel<4, 0>(const StreamParams, const SelectParams, const MultiParams);
}
""")
    asm = tmp_path / "tu.s"
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S",
                           "-I" + os.path.join(ROOT, "approximate-spmv-topk_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                           "-o", str(asm), str(tu)], stderr=subprocess.DEVNULL)
    fn, seq, chains = None, [], {}

    def flush():
        if fn:
            chains[fn] = [len(m.group(0)) // 2 for m in re.finditer(r"(?:WL){6,}", "".join(seq))]

    for ln in asm.read_text().split("\n"):
        if re.match(r"^_ZN6tkspmv\w+:", ln):
            flush()
            fn, seq = ln.split(":")[0], []
        t = ln.strip().split(";")[0].strip()
        if not ln.startswith("\t") or not t:
            continue
        if t.startswith(("global_load", "buffer_load", "flat_load")):
            seq.append("L")
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            seq.append("W")
        elif t.startswith(("global_store", "global_atomic", "s_sleep")):
            seq.append("x")
    flush()
    chains = {n: r for n, r in chains.items() if "select_kernel" not in n and "select_group_kernel" not in n}  # (non-template kernels of the headers)
    assert len(chains) == 6, sorted(chains)
    for name, runs in chains.items():  # (round 5: x is staged with buffer loads -- no clamped addresses --, the exact kernel has no such run either)
        assert len(runs) == 0, (name, runs)
