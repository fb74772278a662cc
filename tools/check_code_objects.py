#!/usr/bin/env python3
"""Build gate: no product kernel may spill registers or use private (scratch) memory.

Reads the AMDGPU metadata notes of every gfx950 code object embedded in build/csrc/*.o (hipcc puts the device code into the
object's .hip_fatbin section as an offload bundle) and lists, per kernel, vgpr_count / vgpr_spill_count / sgpr_spill_count /
private_segment_fixed_size.  Exit code 1 when a kernel spills or has a non-zero private segment, unless it is listed in ALLOWED
with a reason.  Called by __graft_entry__.build() and tests/test_abi_cpu.py.

usage: python tools/check_code_objects.py [--all] [object files ...]
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("P2I_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

# demangled-name substrings of kernels that may keep a private segment, each with the reason (none at the moment)
ALLOWED: dict = {}


def kernels_of(obj: str):
    """The amdhsa.kernels records (dicts with '.name', '.vgpr_count', ...) of the gfx950 code object inside one host object file."""
    import yaml
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat"), os.path.join(td, "co")
        sections = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-S", obj], text=True)
        if ".hip_fatbin" not in sections:            # host-only translation unit (tape.hip): no kernels
            return []
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", f"--targets={TARGET}", f"--input={fat}",
                               f"--output={co}", "--unbundle"])
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
        lines = notes.splitlines()
        start = next(i for i, l in enumerate(lines) if l.strip() == "---")
        end = next((i for i in range(start + 1, len(lines)) if lines[i].strip() == "..."), len(lines))
        meta = yaml.safe_load("\n".join(lines[start + 1:end]))
        recs = [{k.lstrip("."): v for k, v in rec.items() if k != ".args"} for rec in meta.get("amdhsa.kernels", [])]
        # A private segment without spilled VGPRs can be a frame object that no instruction touches (hipcc keeps an emergency slot
        # next to SGPR-to-VGPR-lane spills): count the scratch instructions of such kernels in the disassembly
        sus = [r for r in recs if r.get("private_segment_fixed_size", 0) and not r.get("vgpr_spill_count", 0)]
        if sus:
            dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], text=True)
            cur, counts = None, {}
            for line in dis.splitlines():
                if line.endswith(">:"):
                    cur = line.split("<")[-1][:-2]
                elif cur is not None and ("scratch_" in line or "s[0:3], 0 offen" in line):
                    counts[cur] = counts.get(cur, 0) + 1
            for r in sus:
                r["scratch_instructions"] = counts.get(r["name"], 0)
    return recs


def demangle(names):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            return subprocess.check_output([tool], input="\n".join(names), text=True).splitlines()
        except Exception:
            continue
    return list(names)


def check(objs=None, verbose=False):
    objs = objs or sorted(glob.glob(os.path.join(ROOT, "build", "csrc", "*.o")))
    if not objs:
        raise RuntimeError("no object files under build/csrc: run `make -C p2i-gan-benchmark_amd/csrc` first")
    bad, total = [], 0
    for obj in objs:
        recs = kernels_of(obj)
        names = demangle([r["name"] for r in recs])
        for r, dn in zip(recs, names):
            total += 1
            # (SGPR "spills" are v_writelane / v_readlane into a spare VGPR, not memory: reported with --all, not gated)
            spill = r.get("vgpr_spill_count", 0) or (r.get("private_segment_fixed_size", 0) and r.get("scratch_instructions", 1))
            allowed = next((why for key, why in ALLOWED.items() if key in dn), None)
            if verbose or spill or r.get("private_segment_fixed_size", 0):
                print(f"{os.path.basename(obj):18s} vgpr {r.get('vgpr_count', -1):3d} agpr {r.get('agpr_count', 0):3d} vspill {r.get('vgpr_spill_count', 0):4d} "
                      f"sspill {r.get('sgpr_spill_count', 0):3d} scratch {r.get('private_segment_fixed_size', 0):5d} B  lds {r.get('group_segment_fixed_size', 0):6d}  {dn[:150]}"
                      + (f"   [allowed: {allowed}]" if spill and allowed else "")
                      + ("   [private segment is an untouched frame slot: 0 scratch instructions]" if r.get("scratch_instructions") == 0 else ""))
            if spill and not allowed:
                bad.append(dn)
    return total, bad


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    total, bad = check(args or None, verbose="--all" in sys.argv)
    print(f"check_code_objects: {total} kernels, {len(bad)} with spills / private memory")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
