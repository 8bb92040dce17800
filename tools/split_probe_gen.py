"""Phase stamps for the split-bf16 GEMM: generates an INSTRUMENTED copy of the product kernel (csrc/gemm_split.hip, namespace dsc_split)
and builds tools/split_probe (tools/split_probe.hip = the harness around it).  Nothing is duplicated in the tree: the kernel text is taken
from the product source at build time, s_memtime stamps are inserted at anchor lines (asserted to exist), the namespace is renamed.

    python tools/split_probe_gen.py            # -> tools/_build/split_probe_kernel.inc, tools/split_probe (gfx950 binary, no torch)
    ./tools/split_probe [gn] [res]             # on the GPU box: bit-compare with the product launch, sustained us, cycles per phase

Stamps per K tile and wave (SGPR pairs, stored by lane 0 at the end of the tile so that no wait lands inside the tile):
    0 before the top-of-tile s_waitcnt   1 after the block barrier   2 before the first MFMA block   3.. after token block i
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "diffuscene_amd", "csrc", "gemm_split.hip")
OUT_DIR = os.path.join(ROOT, "tools", "_build")


def replace_once(text, old, new):
    assert text.count(old) == 1, "anchor must occur exactly once: %r (found %d)" % (old, text.count(old))
    return text.replace(old, new)


def generate():
    src = open(SRC).read()
    a = src.index("namespace dsc_split {")
    b = src.index("}  // namespace dsc_split") + len("}  // namespace dsc_split")
    k = src[a:b].replace("namespace dsc_split", "namespace dsc_split_probe")
    if "--ordered" in sys.argv:
        # fragment reads at the top of a tile in the order the first MFMAs need them (x1 + the four w3 fragments, then x3 + w1, then
        # x2 + w2), fenced, instead of the scheduler's order (the product's first MFMA waits for 12 of the 15 reads)
        k = replace_once(k,
            "        for (int pl = 0; pl < 3; ++pl) xf[0][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * X_PLANE + xoff[0]);\n"
            "#pragma unroll\n        for (int j = 0; j < 4; ++j)\n#pragma unroll\n"
            "            for (int pl = 0; pl < 3; ++pl) wf[j][pl] = *reinterpret_cast<const bf16x8*>(cur + woff[j] + pl * B_PLANE);\n",
            "        for (int o = 0; o < 3; ++o) {\n"
            "            const int xp = o == 0 ? 0 : (o == 1 ? 2 : 1), wp = o == 0 ? 2 : (o == 1 ? 0 : 1);\n"
            "            xf[0][xp] = *reinterpret_cast<const bf16x8*>(cur + xp * X_PLANE + xoff[0]);\n"
            "#pragma unroll\n            for (int j = 0; j < 4; ++j) wf[j][wp] = *reinterpret_cast<const bf16x8*>(cur + woff[j] + wp * B_PLANE);\n"
            "            __builtin_amdgcn_sched_barrier(0);\n        }\n")
    # the rejected role-split form of the K loop (tools/split_probe_roles.inc) as template parameter ROLES of the copy
    k = replace_once(k, "template <bool GN, int WM, int WN, int RB, bool DSPREAD = DSC_SPLIT_DSPREAD>\n__global__",
                     "template <bool GN, int WM, int WN, int RB, bool ROLES = false, bool DSPREAD = DSC_SPLIT_DSPREAD>\n__global__")
    roles = "".join(l for l in open(os.path.join(ROOT, "tools", "split_probe_roles.inc")) if not l.startswith("//"))
    k = replace_once(k, "    for (int u = 0; u < NIT; ++u) store_item(u, smem);\n    for (int kt = 0; kt < KT; ++kt) {",
                     "    for (int u = 0; u < NIT; ++u) store_item(u, smem);\n" + roles + "    for (int kt = 0; kt < KT; ++kt) {")
    k = replace_once(k, "template <bool GN, int WM, int WN, int RB>\nint launch(", "template <bool GN, int WM, int WN, int RB, bool ROLES = false>\nint launch(")
    k = replace_once(k, "hipLaunchKernelGGL((gemm_split_kernel<GN, WM, WN, RB>),", "hipLaunchKernelGGL((gemm_split_kernel<GN, WM, WN, RB, ROLES>),")
    k = replace_once(k, "void gemm_split_kernel(const dsc_gemm_args p, const int ntok) {",
                     "void gemm_split_kernel(const dsc_gemm_args p, const int ntok, unsigned long long* const stamps, const int stamp_stride) {\n"
                     "    PROBE_DECL")
    k = replace_once(k, "        __builtin_amdgcn_s_waitcnt(0x0070);              // vmcnt(0) lgkmcnt(0)",
                     "        PROBE_STAMP(0)\n        __builtin_amdgcn_s_waitcnt(0x0070);              // vmcnt(0) lgkmcnt(0)")
    k = replace_once(k, "        __syncthreads();                                 // everyone's are; nobody reads the other stage any more",
                     "        __syncthreads();\n        PROBE_STAMP(1)")
    # the sched_barrier right in front of the token-block loop
    k = replace_once(k, "        __builtin_amdgcn_sched_barrier(0);\n#pragma unroll\n        for (int i = 0; i < RB; ++i) {",
                     "        __builtin_amdgcn_sched_barrier(0);\n        PROBE_STAMP(2)\n#pragma unroll\n        for (int i = 0; i < RB; ++i) {")
    # end of a token block = the sched_barrier that closes it
    k = replace_once(k, "            __builtin_amdgcn_sched_barrier(0);\n        }\n    }\n    __builtin_amdgcn_s_waitcnt(0x0f70);",
                     "            __builtin_amdgcn_sched_barrier(0);\n            PROBE_STAMP_BLOCK(i)\n        }\n        PROBE_FLUSH\n    }\n"
                     "    __builtin_amdgcn_s_waitcnt(0x0f70);\n    PROBE_END")
    # the role-split K loop (ROLES = true) carries the same stamps
    k = replace_once(k, "            __builtin_amdgcn_s_waitcnt(0x0070);\n            __syncthreads();\n",
                     "            PROBE_STAMP(0)\n            __builtin_amdgcn_s_waitcnt(0x0070);\n            __syncthreads();\n            PROBE_STAMP(1)\n")
    k = replace_once(k, "            __builtin_amdgcn_sched_barrier(0);\n            bf16x8 wf[4][3], xf[2][3];",
                     "            __builtin_amdgcn_sched_barrier(0);\n            PROBE_STAMP(2)\n            bf16x8 wf[4][3], xf[2][3];")
    k = replace_once(k, "                __builtin_amdgcn_sched_barrier(0);\n            }\n            if (first) {\n                dma_tile(kn, nxt);\n"
                        "#pragma unroll\n                for (int u = 0; u < NIT; ++u) store_item(u, nxt);\n            }\n",
                     "                __builtin_amdgcn_sched_barrier(0);\n                PROBE_STAMP_BLOCK(i)\n            }\n            if (first) {\n"
                     "                dma_tile(kn, nxt);\n#pragma unroll\n                for (int u = 0; u < NIT; ++u) store_item(u, nxt);\n            }\n"
                     "            PROBE_FLUSH\n")
    k = replace_once(k, "    hipLaunchKernelGGL((gemm_split_kernel<GN, WM, WN, RB, ROLES>), dim3(grid, (unsigned)a->batch), dim3(64 * WM * WN), 0, s, *a, ntok);",
                     "    hipLaunchKernelGGL((gemm_split_kernel<GN, WM, WN, RB, ROLES>), dim3(grid, (unsigned)a->batch), dim3(64 * WM * WN), 0, s, *a, ntok, "
                     "g_stamps, g_stamp_stride);")
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, "split_probe_kernel.inc"), "w") as f:
        f.write("// GENERATED by tools/split_probe_gen.py from diffuscene_amd/csrc/gemm_split.hip -- do not edit\n" + k + "\n")


def build():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = os.path.join(ROOT, "tools", "split_probe")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "diffuscene_amd", "csrc"), "-I", OUT_DIR, os.path.join(ROOT, "tools", "split_probe.hip"), "-o", out,
           "-L", os.path.join(ROOT, "diffuscene_amd"), "-ldiffuscene_hip", "-Wl,-rpath,$ORIGIN/../diffuscene_amd"]
    cmd += [a for a in sys.argv[1:] if a.startswith("-D")]
    for a in sys.argv[1:]:
        if a.startswith("--out="):
            cmd[cmd.index("-o") + 1] = out = os.path.join(ROOT, "tools", a[6:])
    subprocess.check_call(cmd)
    print("built", out)


if __name__ == "__main__":
    generate()
    build()
