"""Register / spill metadata of the hot kernels, read from the compiler's own output (hipcc -S for gfx950, no GPU needed):
the dominant instantiations must not spill, the overlap-friendly launches must fit the registers four scan waves leave on
a SIMD, and no hot loop of the Foveal prefix-sum scan may touch scratch memory."""
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

from shadowing_amd import _build

CSRC = _build.CSRC


def _asm(name: str, tmp: Path) -> str:
    out = tmp / (name + ".s")
    flags = [f for f in _build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    res = subprocess.run([_build.hipcc_path(), *flags, f"-I{_build.INCLUDE}", f"-I{CSRC}", "-S", "--cuda-device-only",
                          str(CSRC / (name + ".hip")), "-o", str(out)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    return out.read_text()


def _kernels(txt: str) -> dict:
    meta = {}
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)   # noqa: E731
        meta[g("name")] = dict(vgpr=int(g("vgpr_count")), spill=int(g("vgpr_spill_count")), scratch=int(g("private_segment_fixed_size")))
    return meta


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("isa")
    names = ["psh_stream", "psh_fused", "psh_embed_px", "psh_embed_mx", "psh_lq"]
    with ThreadPoolExecutor(max_workers=5) as pool:
        return dict(zip(names, pool.map(lambda n: _asm(n, tmp), names)))


def test_overlap_launches_fit_beside_each_other(asm):
    """Four scan waves of 112 VGPRs leave 64 of a SIMD's 512: the sample and ranking kernels must need no more, the scan no
    more than 112, none of the W = 20 instantiations may spill."""
    k = _kernels(asm["psh_stream"])
    scan = {n: m for n, m in k.items() if "stream_scan_kernel" in n}
    # (the one-wave sample blocks: "ELi1EE"; the four-wave form serves two- and three-query steps, which run alone)
    small = {n: m for n, m in k.items() if ("stream_sample_kernel" in n and n.endswith("ELi1EEEvNS_8ScanArgsENS_9FusedArgsE")) or "stream_rank_kernel" in n}
    assert scan and small
    assert all(m["vgpr"] <= 112 for m in scan.values()), scan
    assert all(m["vgpr"] <= 64 and m["spill"] == 0 and m["scratch"] == 0 for m in small.values()), small
    assert all(m["spill"] == 0 for n, m in scan.items() if "ILi20E" in n), scan


def test_fused_launch_does_not_spill(asm):
    k = _kernels(asm["psh_fused"])
    fused = {n: m for n, m in k.items() if "scan_fused_kernel" in n}
    assert len(fused) == 16 and all(m["vgpr"] <= 128 for m in fused.values()), fused
    # <WT, ALIGNED, HINTED, BLK>: the launches psh_scan_topk issues (BLK = false) compile exactly as before the blocking entry
    # existed -- no spill anywhere; the blocking caller's launches (BLK: path gather + completion words in the ranking phase)
    # do not spill in the aligned forms (16-byte aligned rows: the benchmark's), the unaligned sampled ones keep ONE value
    # (the thread index, stored at the start and read back outside the scan loop) in scratch
    plain = {n: m for n, m in fused.items() if n.endswith("ELb0EEEvNS_8ScanArgsENS_9FusedArgsE")}
    blk = {n: m for n, m in fused.items() if n.endswith("ELb1EEEvNS_8ScanArgsENS_9FusedArgsE")}
    assert len(plain) == 8 and len(blk) == 8
    assert all(m["spill"] == 0 for m in plain.values()), plain
    assert all(m["spill"] == 0 for n, m in blk.items() if re.search(r"ILi\d+ELb1ELb[01]ELb1E", n)), blk
    assert all(m["spill"] <= 1 for m in blk.values()), blk


def test_foveal_prefix_sum_scan_keeps_scratch_out_of_its_hot_loops(asm):
    """embed_px_kernel: the sample instantiations do not spill at all; the full-scan ones keep a dozen values (set-up,
    verification of survivors) in scratch but NOT inside the row loops (every loop with >= 16 packed fma)."""
    txt = asm["psh_embed_px"]
    k = _kernels(txt)
    boot = {n: m for n, m in k.items() if "embed_px_kernelILb" in n and "ELi0ELi1024" in n}
    assert boot and all(m["spill"] == 0 for m in boot.values()), boot
    for name in [n for n in k if "embed_px_kernelILb" in n and "ELi1ELi1024" in n]:
        body = txt[txt.index("\n" + name + ":"):]
        body = body[:body.index("s_endpgm")].splitlines()
        labels = {m.group(1): i for i, ln in enumerate(body) for m in [re.match(r"(\.LBB\d+_\d+):", ln)] if m}
        hot = 0
        for i, ln in enumerate(body):
            m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", ln)
            if m and labels.get(m.group(1), i + 1) < i:                 # a backward branch: labels[...] .. i is a loop
                loop = body[labels[m.group(1)]:i]
                if sum("v_pk_fma_f32" in x for x in loop) >= 16 and not any(re.match(r"\s+s_cbranch", x) for x in loop[:-1]):
                    hot += 1
                    assert not any("scratch_" in x for x in loop), f"{name}: scratch access inside a row loop"
        assert hot >= 1, f"{name}: no row loop found"


def test_wavelet_scan_on_the_matrix_cores_does_not_spill_and_keeps_its_product_loop_clean(asm):
    """embed_mx_kernel: the bootstrap and the full scan (one product; per-query pass on the matrix cores or not) do not spill
    for 1, 2 or 3 row groups (the split-product full scan, an A/B flag, may); the production product loop (three row groups)
    is MFMAs and 16-byte LDS reads only -- 48 + 28 per turn, no vector-ALU instruction between them."""
    txt = asm["psh_embed_mx"]
    k = {n: m for n, m in _kernels(txt).items() if "embed_mx_kernel" in n}
    assert len(k) == 24
    prod = {n: m for n, m in k.items() if not re.search(r"ELi1ELi[123]ELi3ELb0E", n)}     # (FILTER, NP = 3: PSH_FLAG_EMBED_MX_SPLIT)
    assert len(prod) == 18 and all(m["spill"] == 0 and m["scratch"] == 0 for m in prod.values()), prod
    for name in [n for n in k if re.search(r"ILb[01]ELi1ELi3ELi1ELb1E", n)]:
        body = txt[txt.index("\n" + name + ":"):]
        body = body[:body.index("s_endpgm")].splitlines()
        loops = []
        labels = {m.group(1): i for i, ln in enumerate(body) for m in [re.match(r"(\.LBB\d+_\d+):", ln)] if m}
        for i, ln in enumerate(body):
            m = re.match(r"\s+s_c?branch\w* (\.LBB\d+_\d+)", ln)
            if m and labels.get(m.group(1), i + 1) < i:
                ops = [x.split()[0] for x in body[labels[m.group(1)]:i] if re.match(r"\s+[a-z]", x)]
                if sum(o.startswith("v_mfma") for o in ops) >= 48:
                    loops.append(ops)
        assert loops, name
        ops = min(loops, key=len)                                         # the innermost loop holding the 48 MFMAs
        assert sum(o.startswith("v_mfma") for o in ops) == 48 and sum(o == "ds_read_b128" for o in ops) == 28, ops
        assert not [o for o in ops if o.startswith("v_") and not o.startswith("v_mfma") and o not in ("v_add_u32_e32",)], ops
        assert sum(o == "v_add_u32_e32" for o in ops) <= 2, ops


def test_long_window_scans_do_not_spill(asm):
    """stream_scan_long_kernel (one to three queries: at most 128 registers, four waves a SIMD) and scan_lq_kernel (the batched
    long-window scan: eight waves a CU, 256 registers a lane) keep everything in registers; a scan_lq segment's MFMAs come in
    groups of four independent tiles."""
    k = _kernels(asm["psh_stream"])
    long_k = {n: m for n, m in k.items() if "stream_scan_long_kernel" in n}
    assert len(long_k) == 6 and all(m["spill"] == 0 and m["scratch"] == 0 and m["vgpr"] <= 128 for m in long_k.values()), long_k
    assert all(m["vgpr"] <= 112 for n, m in long_k.items() if "ELi1EEEv" in n), long_k      # one query: room for a sample / ranking wave beside four of these
    lq = {n: m for n, m in _kernels(asm["psh_lq"]).items() if "scan_lq_kernel" in n}
    assert len(lq) == 8 and all(m["spill"] == 0 and m["scratch"] == 0 and m["vgpr"] <= 256 for m in lq.values()), lq
