#!/usr/bin/env python3
"""Where the rollout kernel's vector instructions go, by source function (static count, no GPU needed).

Builds mrsim_kernels.hip with -gline-tables-only and -DMRSIM_BUDGET_BUILD (the general RK45 path traps instead of running,
so the compiler keeps only the common straight-line step inside the time loop), reads the generated ISA of the
flag-specialised DDPG rollout kernel (fp64 carry) and attributes every VALU instruction of the loop body to the function
of mrsim_device.h / the part of the kernel its .loc line belongs to (inlined code keeps the callee's lines).
The auto-reset block (cold in this workload: once per 51 steps) traps in that build as well.
Compare the hot total with SQ_INSTS_VALU per wave-step in profiles/rNN/pmc_valu.json (271)."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mr_rl_amd", "csrc")
KERNEL_FMT = "_ZN5mrsim17mr_rollout_kernelILb1ELi%dELb%dELj720005EEEvNS_7KParamsENS_9StateArgsENS_11RolloutArgsE"
COLD = set()


def function_ranges(path):
    """[(first_line, name)] of the function definitions of a source file, in line order"""
    out = []
    for i, line in enumerate(open(path), 1):
        t = line.strip()
        if not (("__device__" in t or "__global__" in t or t.startswith("static ")) and "(" in t) or t.startswith("//"):
            continue
        if t.endswith(";") and ")" in t and "{" not in t:
            continue  # declaration
        head = t[:t.index("(")]
        head = re.sub(r"__launch_bounds__\s*$", "", head).strip()
        m = re.search(r"(\w+)\s*$", head)
        if m and m.group(1) not in ("__launch_bounds__", "if", "for", "while"):
            out.append((i, m.group(1)))
    return out


def owner(ranges, line):
    name = "?"
    for first, n in ranges:
        if first > line:
            break
        name = n
    return name


def main():
    # usage: isa_budget.py [per_stage|collapsed] [nominal|mismatched] [actor math: f32|bf16x3|bf16 -> the actor-in-the-loop kernel]
    law = sys.argv[1] if len(sys.argv) > 1 else "per_stage"
    mis = (sys.argv[2] if len(sys.argv) > 2 else "nominal") == "mismatched"
    KERNEL = KERNEL_FMT % (4 if law == "collapsed" else 2, 1 if mis else 0)
    actor = sys.argv[3] if len(sys.argv) > 3 else None
    if actor is not None:
        fl, act = {"f32": (7011461, 1), "bf16x3": (40565893, 2), "bf16": (74120325, 3)}[actor]
        KERNEL = ("_ZN5mrsim26mr_rollout_actor_fl_kernelILb1ELi%dELb%dELj%dELi%dEEEvNS_7KParamsENS_9StateArgsENS_11RolloutArgsENS_9ActorArgsE"
                  % (4 if law == "collapsed" else 2, 1 if mis else 0, fl, act))
    tmp = tempfile.mkdtemp(prefix="isa_budget_")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-fno-slp-vectorize", "-gline-tables-only", "-DMRSIM_BUDGET_BUILD=1", "-shared", "-I../../include",
           "-save-temps=obj", "-o", os.path.join(tmp, "lib.so"), "mrsim_kernels.hip"]
    subprocess.run(cmd, cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(tmp, "mrsim_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    files = {int(m.group(1)): m.group(2) for m in re.finditer(r'^\s*\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]+)"', asm, re.M)}
    body = asm[asm.index(KERNEL + ":"):]
    body = body[:body.index(".Lfunc_end")]
    ranges = {"mrsim_device.h": function_ranges(os.path.join(CSRC, "mrsim_device.h")),
              "mrsim_actor.h": function_ranges(os.path.join(CSRC, "mrsim_actor.h")),
              "mrsim_kernels.hip": function_ranges(os.path.join(CSRC, "mrsim_kernels.hip"))}
    # the time loop = the blocks the compiler marks "in Loop:"
    cur, in_loop = ("?", 0), False
    valu, other = collections.Counter(), collections.Counter()
    for line in body.split("\n"):
        t = line.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur = (os.path.basename(files.get(int(m.group(1)), "?")), int(m.group(2)))
            continue
        if re.match(r"\.LBB\d+_\d+:", t):
            in_loop = "in Loop" in t or "Loop Header" in t
            continue
        if "in Loop" in t and t.startswith(";"):
            in_loop = True
            continue
        if not t or t[0] in ".;" or not in_loop:
            continue
        op = t.split()[0]
        f, ln = cur
        name = owner(ranges[f], ln) if f in ranges else f
        if f == "mrsim_kernels.hip":
            name = "kernel body (loads, stores, loop)"
        (valu if op.startswith("v_") else other)[name] += 1
    hot = sum(v for k, v in valu.items() if k not in COLD)
    print("VALU instructions inside the time loop of the DDPG rollout kernel, by source function (common path); noise law %s, %s model" % (law, "mismatched" if mis else "nominal"))
    for k, v in sorted(valu.items(), key=lambda kv: -kv[1]):
        print("  %-38s %4d%s   (+%d scalar / memory)" % (k, v, "  [cold: auto-reset]" if k in COLD else "", other.get(k, 0)))
    print("  %-38s %4d   (rocprofv3 SQ_INSTS_VALU per wave-step: see profiles/*/pmc_valu.json)" % ("hot total", hot))


if __name__ == "__main__":
    main()
