#!/bin/bash
# Dev tool: where the time inside the plan's first launch goes.  Builds build_tools/libmalcs_clocks.so, a variant of the
# library whose k_plan_ends stamps wall_clock64 (100 MHz) per workgroup (start, end) and per phase of the first sample
# workgroup into the sample's distance buffer, which malcs_batch_plan_dump(9) of that variant hands out; the tree is
# left as it was.  On the GPU box:  MALCS_LIB=$PWD/build_tools/libmalcs_clocks.so python tools/plan_clocks.py
set -e
cd "$(dirname "$0")/../algebra_amd/csrc"
cp malcs_plan.inc /tmp/plan_keep.inc; cp malcs_diag.hip /tmp/diag_keep.hip
trap 'cp /tmp/plan_keep.inc malcs_plan.inc; cp /tmp/diag_keep.hip malcs_diag.hip' EXIT
python3 - <<'PY'
p = 'malcs_plan.inc'; s = open(p).read()
old = "    if (SAMPLE == 2 && blockIdx.x < A.sample_blocks) {\n"
new = '''    const unsigned long long clk0 = wall_clock64();
    struct stamp { const plan_args &A; unsigned long long t0; __device__ ~stamp() { if (threadIdx.x == 0) { uint32_t *o = reinterpret_cast<uint32_t *>(A.sample_dist); o[2 * blockIdx.x] = (uint32_t)t0; o[2 * blockIdx.x + 1] = (uint32_t)wall_clock64(); } } } stamp_{A, clk0};
''' + old + "        if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<uint32_t *>(A.sample_dist)[1200] = (uint32_t)clk0;\n"
assert old in s; s = s.replace(old, new, 1)
def ins(after, k):
    global s
    assert after in s, after
    s = s.replace(after, after + f"\n        if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<uint32_t *>(A.sample_dist)[1200 + {k}] = (uint32_t)wall_clock64();", 1)
ins("make_uint4(0u, 0u, 0u, 0u);\n        __syncthreads();", 1)
ins("                if (w < CW) { lds_seq[j * 2u * CW + w] = va[i]; lds_seq[j * 2u * CW + CW + w] = vb[i]; }\n            }\n        }\n        __syncthreads();", 2)
ins("        uint32_t ca = __shfl_up(ea, 1, 64), cb = __shfl_up(eb, 1, 64);           // ... and starts: where its neighbour's ends", 3)
ins("        const int32_t d = llcs_band32_lane<NP>(lds_seq, ao + (ca / 32u) * NP, m, bo, (lb + 31u) >> 5, cb, n, valid);", 4)
open(p, 'w').write(s)
p = 'malcs_diag.hip'; s = open(p).read()
old = '    default: malcs_set_error("no such plan array"); return MALCS_ERR_ARG;'
new = "    case 9: src = reinterpret_cast<const uint32_t *>(d.d_sample_dist); n = (uint64_t)SAMPLE_SLOTS * d.n_sample + 64; break;\n" + old
assert old in s; s = s.replace(old, new, 1)
open(p, 'w').write(s)
PY
mkdir -p ../../build_tools
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-value -c malcs_diag.hip -o /tmp/diag_clocks.o
objs=$(ls build/*.o | grep -v 'build/malcs_diag.o')
hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_tools/libmalcs_clocks.so $objs /tmp/diag_clocks.o -ldl
echo "built build_tools/libmalcs_clocks.so"
