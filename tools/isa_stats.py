#!/usr/bin/env python3
"""ISA accounting of the shipped gfx950 code object: per kernel, the instruction mix that prices the field arithmetic.

    python tools/isa_stats.py [--so spartan-bn254_amd/libsbn254_hip.so] [--kernels k_acc_first,k_comb_rows,...] [--json out.json]

Extracts the gfx950 code object from the .so (llvm-objdump --offloading), reads the kernel descriptors' metadata
(llvm-readelf --notes: VGPRs, SGPRs, scratch, LDS) and disassembles it (llvm-objdump -d --mcpu=gfx950), then counts per kernel:
  v_mad_u64_u32   (+ v_mad_i64_i32) the only wide multiplier of the CDNA4 VALU: 4.2 cycles per wave-instruction (tools/micro/ibench.hip)
  v_mul_lo_u32    the Montgomery quotient digits (4.2)
  other VALU      shifts, masks, limb additions, selects, moves: 2.3 cycles for plain VOP2 forms, 4.2 for VOP3 / 64-bit / carry forms
  s_nop, SALU, VMEM (global/buffer/scratch), LDS, waitcnt
and prices them roughly: cycles ~ 4.2 * (mad + mul_lo) + 3 * other VALU; `mult_share` = the multiplier's part of that (1.0 = nothing
but products).  Needs no GPU (runs on the build machine)."""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAD_CYC, VALU_CYC = 4.2, 3.0


def extract(so):
    tmp = tempfile.mkdtemp(prefix="isa_")
    dst = os.path.join(tmp, os.path.basename(so))
    shutil.copy(so, dst)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", dst], check=True, stdout=subprocess.DEVNULL, cwd=tmp)
    cos = [os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f]
    if not cos:
        raise SystemExit(f"no gfx950 code object inside {so}")
    return tmp, cos[0]


def metadata(co):
    """kernel symbol -> descriptor metadata.  A kernel's record in the amdhsa.kernels list starts at its first key (`- .agpr_count:`) and
    its keys come in alphabetical order, so .symbol precedes .vgpr_count: records are closed at the next record's start."""
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out = {}
    cur = None
    keep = ("vgpr_count", "sgpr_count", "agpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size")

    def flush():
        if cur and cur.get("symbol", "").endswith(".kd"):
            out[cur["symbol"][:-3]] = cur

    for line in txt.splitlines():
        m = re.match(r"^(\s*)(- )?\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        dash, k, v = m.group(2), m.group(3), m.group(4).strip()
        if dash and k == "agpr_count":
            flush()
            cur = {}
        if cur is None:
            continue
        if k in keep:
            cur[k] = int(v)
        elif k == "symbol":
            cur["symbol"] = v.strip("'\"")
    flush()
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    d = p.stdout.splitlines() if p.returncode == 0 else names
    return dict(zip(names, d))


def disasm_counts(co):
    p = subprocess.Popen([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", "--no-show-raw-insn", co], stdout=subprocess.PIPE, text=True)
    counts = {}
    cur = None
    for line in p.stdout:
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = m.group(1)
            counts[cur] = {"total": 0, "v_mad_u64_u32": 0, "v_mul_lo_u32": 0, "other_valu": 0, "s_nop": 0, "salu": 0, "vmem": 0, "scratch": 0, "lds": 0, "waitcnt": 0}
            continue
        if cur is None:
            continue
        t = line.strip().split()
        if not t:
            continue
        op = t[0]
        if not re.match(r"^[a-z]", op) or op.startswith("//"):
            continue
        c = counts[cur]
        c["total"] += 1
        if op in ("v_mad_u64_u32", "v_mad_i64_i32"):          # the wide multiplier, unsigned and signed form
            c["v_mad_u64_u32"] += 1
        elif op == "v_mul_lo_u32":
            c["v_mul_lo_u32"] += 1
        elif op.startswith("v_"):
            c["other_valu"] += 1
        elif op == "s_nop":
            c["s_nop"] += 1
        elif op == "s_waitcnt":
            c["waitcnt"] += 1
        elif op.startswith("scratch_"):
            c["scratch"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    p.wait()
    return counts


def waves_per_simd(vgprs):
    alloc = -(-max(vgprs, 1) // 8) * 8
    return min(8, 512 // alloc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "spartan-bn254_amd", "libsbn254_hip.so"))
    ap.add_argument("--kernels", default="k_acc_first,k_acc_extra,k_comb_rows,k_reduce_l1,k_reduce_combine,k_sc_eval,k_sc_bind_eval,k_sc_round,k_bind_top,k_dot,k_digits_store,k_sort_rows,k_scatter_lds",
                    help="comma-separated substrings of the (demangled) kernel names to report; 'all' for every kernel")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    tmp, co = extract(args.so)
    try:
        meta = metadata(co)
        cnt = disasm_counts(co)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    names = sorted(meta.keys())
    dm = demangle(names)
    want = None if args.kernels == "all" else [w for w in args.kernels.split(",") if w]
    rows = []
    for n in names:
        pretty = re.sub(r"^void ", "", dm.get(n, n))
        short = re.sub(r"\(.*$", "", pretty)
        if want and not any(w in short for w in want):
            continue
        c = cnt.get(n)
        if not c:
            continue
        m = meta[n]
        mult = c["v_mad_u64_u32"] + c["v_mul_lo_u32"]
        cyc_mult = MAD_CYC * mult
        cyc_other = VALU_CYC * c["other_valu"]
        rows.append({"kernel": short, "vgpr": m.get("vgpr_count"), "agpr": m.get("agpr_count", 0), "sgpr": m.get("sgpr_count"), "scratch_B": m.get("private_segment_fixed_size"),
                     "lds_B": m.get("group_segment_fixed_size"), "waves_per_simd": waves_per_simd((m.get("vgpr_count") or 0) + (m.get("agpr_count") or 0)),
                     **c, "mult_share": round(cyc_mult / (cyc_mult + cyc_other), 3) if (cyc_mult + cyc_other) else None,
                     "products_equiv": round(c["v_mad_u64_u32"] / 162.0, 1)})
    hdr = f"{'kernel':44s} {'vgpr':>4s} {'w/S':>3s} {'scr':>4s} {'instr':>6s} {'mad64':>6s} {'mullo':>5s} {'oVALU':>6s} {'s_nop':>5s} {'vmem':>4s} {'lds':>4s} {'mult%':>6s}"
    print(hdr)
    for r in rows:
        print(f"{r['kernel'][:44]:44s} {r['vgpr']:4d} {r['waves_per_simd']:3d} {r['scratch_B']:4d} {r['total']:6d} {r['v_mad_u64_u32']:6d} {r['v_mul_lo_u32']:5d} {r['other_valu']:6d} {r['s_nop']:5d} {r['vmem'] + r['scratch']:4d} {r['lds']:4d} {100 * (r['mult_share'] or 0):6.1f}")
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
