#!/usr/bin/env python3
"""Post-build ISA checks and per-kernel resource report for libcovgram.so's translation units (no GPU needed: hipcc -S --cuda-device-only).

  python tools/check_isa.py report  <tu.hip> [-DCOVGRAM_FAM=6 ...] [--filter SUBSTR]   per kernel: VGPRs, AGPRs, SGPRs, scratch bytes, LDS, occupancy,
                                                                                      and the mix of the inner loops (v_pk_*, transcendental, MFMA, ...)
  python tools/check_isa.py lint                                                       the asserted properties (tests/test_host.py runs this):
      1. pack.hpp last_arrival: every kernel that takes a ticket has `s_waitcnt vmcnt(0)` in front of the s_barrier that precedes the
         global_atomic_add (a workgroup-scope release fence emits no vmcnt wait on gfx950)
      2. v_fmac_f64_dpp (grad_bcast.hpp / dense_bcast.hpp inline asm, invisible to the hazard recogniser): no VALU write of the DPP source
         in the 2 instructions before it and no EXEC write in the 5 before it; those kernels use no scratch
      3. the one-pass Sum kernels and the packed MaternP kernels: zero scratch
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "covariancefunctions.jl_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only"]
MFMA = ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"]


def compile_asm(tu: str, extra: list[str]) -> str:
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [HIPCC] + BASE + extra + [os.path.join(CSRC, tu) if not os.path.isabs(tu) else tu, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    with open(out) as f:
        txt = f.read()
    os.unlink(out)
    return txt


def demangle(names: list[str]) -> dict[str, str]:
    if not names:
        return {}
    filt = "c++filt"
    r = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True)
    outs = r.stdout.splitlines()
    return dict(zip(names, outs)) if len(outs) == len(names) else {n: n for n in names}


def kernels(asm: str) -> dict[str, dict]:
    """{mangled: {"body": [instruction lines], "meta": {...}}} from one device assembly file"""
    res: dict[str, dict] = {}
    # bodies: from "<name>:" to the matching ".Lfunc_end"
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M):
        name, body = m.group(1), m.group(2)
        raw = [ln.split(";")[0].strip() for ln in body.splitlines()]
        lines = [ln for ln in raw if ln and not ln.startswith((".", "//")) and not ln.endswith(":")]
        res[name] = {"body": lines, "meta": {}, "raw": [ln for ln in raw if ln]}
    for m in re.finditer(r"\.amdhsa_kernel (\w+)\n(.*?)\.end_amdhsa_kernel", asm, re.S):
        name, blk = m.group(1), m.group(2)
        meta = {}
        for key in ("next_free_vgpr", "next_free_sgpr", "accum_offset", "private_segment_fixed_size", "group_segment_fixed_size"):
            mm = re.search(r"\.amdhsa_" + key + r"\s+(\d+)", blk)
            if mm:
                meta[key] = int(mm.group(1))
        if name in res:
            res[name]["meta"] = meta
    return res


TRANS = ("v_exp_f32", "v_log_f32", "v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32")


def mix(lines: list[str]) -> dict[str, int]:
    c = {"pk": 0, "trans": 0, "mfma": 0, "valu": 0, "salu": 0, "vmem": 0, "lds": 0, "branch": 0, "scratch": 0}
    for ln in lines:
        op = ln.split()[0]
        if op.startswith("v_pk_"):
            c["pk"] += 1
        elif op.startswith(TRANS):
            c["trans"] += 1
        elif op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            c["branch"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("scratch_"):
            c["scratch"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
    return c


def tile_loop_scratch(raw: list[str]) -> tuple[int, int]:
    """(number of single-block loops that hold MFMAs — the tile loops —, scratch instructions inside them)"""
    blocks: list[tuple[str, list[str]]] = []
    cur: tuple[str, list[str]] = ("entry", [])
    for ln in raw:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            blocks.append(cur)
            cur = (m.group(1), [])
        elif not ln.startswith("."):
            cur[1].append(ln)
    blocks.append(cur)
    loops = spills = 0
    for label, b in blocks:
        if any(x.startswith("v_mfma") for x in b) and any(x.startswith("s_cbranch") and x.split()[-1] == label for x in b):
            loops += 1
            spills += sum(1 for x in b if x.startswith("scratch_"))
    return loops, spills


def report(tu: str, extra: list[str], filt: str | None) -> None:
    asm = compile_asm(tu, extra)
    ks = kernels(asm)
    dm = demangle(list(ks))
    for name, k in sorted(ks.items(), key=lambda kv: dm[kv[0]]):
        pretty = dm[name]
        if filt and filt not in pretty:
            continue
        if not k["meta"]:
            continue
        m = k["meta"]
        alloc = (m.get("next_free_vgpr", 0) + 7) // 8 * 8
        occ = min(8, 512 // max(alloc, 1))
        mx = mix(k["body"])
        print(f"{pretty[:150]}\n    vgpr+agpr {m.get('next_free_vgpr')} (alloc {alloc}, {occ} waves/SIMD by registers)  sgpr {m.get('next_free_sgpr')}  "
              f"scratch {m.get('private_segment_fixed_size')} B  lds {m.get('group_segment_fixed_size')} B\n    whole body: {mx}")


def lint() -> int:
    bad = 0
    # every translation unit the checks read, compiled side by side (hipcc is single-threaded per TU)
    from concurrent.futures import ThreadPoolExecutor
    jobs = [("dense_mfma.hip", tuple(MFMA)), ("lowrank.hip", ()), ("grad_fam.hip", ("-DCOVGRAM_FAM=0",)), ("dense_fam.hip", ("-DCOVGRAM_FAM=0",)),
            ("mfma_fam.hip", tuple(MFMA + ["-DCOVGRAM_FAM=6"])), ("mfma_fam.hip", tuple(MFMA + ["-DCOVGRAM_FAM=13"]))]
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        cache = dict(zip(jobs, ex.map(lambda j: compile_asm(j[0], list(j[1])), jobs)))
    compiled = lambda tu, extra: cache[(tu, tuple(extra))]

    def fail(msg):
        nonlocal bad
        bad += 1
        print("FAIL:", msg)

    # 1. ticketed kernels: vmcnt(0) in front of the barrier in front of the ticket's atomic add
    for tu, extra in (("dense_mfma.hip", MFMA), ("lowrank.hip", [])):
        ks = kernels(compiled(tu, extra))
        dm = demangle(list(ks))
        seen = 0
        for name, k in ks.items():
            body = k["body"]
            for i, ln in enumerate(body):
                if not ln.startswith("global_atomic_add") or "sc0" not in ln and "glc" not in ln:   # the returning ticket add
                    if not ln.startswith("global_atomic_add_u32") and not ln.startswith("global_atomic_add "):
                        continue
                # walk back to the nearest s_barrier, then require a vmcnt(0) wait before it with no VMEM store in between
                j = i - 1
                while j >= 0 and not body[j].startswith("s_barrier"):
                    j -= 1
                if j < 0:
                    continue
                seen += 1
                q = j - 1
                ok = False
                while q >= 0 and j - q < 12:
                    if body[q].startswith("s_waitcnt") and "vmcnt(0)" in body[q]:
                        ok = True
                        break
                    if body[q].startswith(("global_store", "buffer_store")):
                        break
                    q -= 1
                if not ok:
                    fail(f"{tu}: {dm[name][:120]}: no `s_waitcnt vmcnt(0)` in front of the ticket's barrier")
        if seen == 0:
            fail(f"{tu}: found no ticketed kernel to check (the pattern matcher is stale)")
        else:
            print(f"ok: {tu}: {seen} ticket sites drain vmcnt before their barrier")

    # 2. DPP fmac hazards in the broadcast kernels
    VALU_DEF = re.compile(r"^v_\w+\s+(v\[\d+:\d+\]|v\d+)")
    for fam, tu in ((0, "grad_fam.hip"), (0, "dense_fam.hip")):
        ks = kernels(compiled(tu, [f"-DCOVGRAM_FAM={fam}"]))
        dm = demangle(list(ks))
        nd = 0
        for name, k in ks.items():
            body = k["body"]
            has = False
            for i, ln in enumerate(body):
                if not ln.startswith("v_fmac_f64_dpp"):
                    continue
                has = True
                nd += 1
                ops = re.findall(r"v\[(\d+):(\d+)\]", ln)
                if len(ops) < 2:
                    continue
                src = set(range(int(ops[1][0]), int(ops[1][1]) + 1))      # src0 of the DPP: the second register operand
                for back in (1, 2):
                    if i - back < 0:
                        break
                    p = body[i - back]
                    if p.startswith("v_fmac_f64_dpp"):
                        continue
                    mm = VALU_DEF.match(p)
                    if mm:
                        r = mm.group(1)
                        regs = set(range(int(r[2:-1].split(":")[0]), int(r[2:-1].split(":")[1]) + 1)) if r.startswith("v[") else {int(r[1:])}
                        if regs & src:
                            fail(f"{tu}: {dm[name][:100]}: VALU write of the DPP source {back} instruction(s) before v_fmac_f64_dpp: `{p}`")
                for back in range(1, 6):
                    if i - back < 0:
                        break
                    p = body[i - back]
                    if re.match(r"^(s_\w+\s+exec|v_cmpx|s_\w+saveexec)", p):
                        fail(f"{tu}: {dm[name][:100]}: EXEC write {back} instruction(s) before v_fmac_f64_dpp: `{p}`")
            if has and k["meta"].get("private_segment_fixed_size", 0) != 0:
                fail(f"{tu}: {dm[name][:100]}: a DPP broadcast kernel uses scratch ({k['meta']['private_segment_fixed_size']} B)")
        if nd == 0:
            fail(f"{tu}: found no v_fmac_f64_dpp (the pattern matcher is stale)")
        else:
            print(f"ok: {tu} (family {fam}): {nd} v_fmac_f64_dpp sites clear of VALU-def / EXEC hazards, no scratch")

    # 3. zero scratch in the packed-profile / one-pass Sum matrix-core kernels
    for fam in (6, 13):
        ks = kernels(compiled("mfma_fam.hip", MFMA + [f"-DCOVGRAM_FAM={fam}"]))
        dm = demangle(list(ks))
        n = n2 = 0
        for name, k in ks.items():
            if not k["meta"]:
                continue
            n += 1
            sc = k["meta"].get("private_segment_fixed_size", 0)
            if "dense_mfma_sym2_kernel" in dm[name]:
                # the two-row-tile symmetric kernel runs at the 168-register limit of 3 waves per SIMD: a few per-stage values (DMA addresses, slab
                # offsets) live in scratch BETWEEN the tile loops (<= 96 B with MaternP's order a compile-time constant); inside the tile loops: none
                ord_ct = re.search(r"dense_mfma_sym2_kernel<\d+, \d+, \d+, (\d+), \d+>", dm[name])
                loops, spills = tile_loop_scratch(k["raw"])
                n2 += 1
                if ord_ct and int(ord_ct.group(1)) >= 1 and (spills != 0 or sc > 96 or loops == 0):
                    fail(f"mfma_fam.hip family {fam}: {dm[name][:120]}: {spills} scratch instructions in {loops} tile loops, scratch {sc} B")
                continue
            if sc != 0:
                fail(f"mfma_fam.hip family {fam}: {dm[name][:120]}: scratch {sc} B")
        print(f"ok: mfma_fam.hip family {fam}: {n - n2} kernels without scratch; {n2} two-row-tile symmetric instances: none inside their tile loops")
    return 1 if bad else 0


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "lint":
        sys.exit(lint())
    if len(sys.argv) >= 3 and sys.argv[1] == "report":
        args = sys.argv[3:]
        filt = None
        if "--filter" in args:
            i = args.index("--filter")
            filt = args[i + 1]
            args = args[:i] + args[i + 2:]
        tu = sys.argv[2]
        extra = list(args)
        if "mfma" in tu:
            extra = MFMA + extra
        report(tu, extra, filt)
        sys.exit(0)
    print(__doc__)
    sys.exit(2)
