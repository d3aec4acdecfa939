#!/bin/bash
# Dev tool: where a wave of k_graph_build spends its time.  Builds build_tools/libmalcs_gbclk.so, a variant of the library
# whose lane 0 stamps wall_clock64 (100 MHz) after the construction, the reachability count, the cursor's atomic and the
# copy-out into a static device array that malcs_dev_gb_clocks() of that variant hands out; the tree is left as it was.
# On the GPU box:  MALCS_LIB=$PWD/build_tools/libmalcs_gbclk.so python tools/gbuild_clocks.py
set -e
cd "$(dirname "$0")/../algebra_amd/csrc"
cp malcs_gbuild.hip /tmp/gbuild_keep.hip
trap 'cp /tmp/gbuild_keep.hip malcs_gbuild.hip' EXIT
python3 - <<'PY'
p='malcs_gbuild.hip'; s=open(p).read()
def rep(old, new):
    global s
    assert s.count(old) >= 1, old
    s = s.replace(old, new, 1)
rep('namespace {\n','__device__ uint32_t gb_stamps[5 * 131072];\nnamespace {\n')
rep('''    if (lane == 0) {
        malcs_core::arena A;''','''    if (lane == 0) {
        const unsigned long long c0 = wall_clock64();
        unsigned long long c2 = 0, c3 = 0;
        malcs_core::arena A;''')
rep('''        int32_t status = A.overflow ? 1 : 0, count = 0, R16 = 0;''','''        const unsigned long long c1 = wall_clock64();
        int32_t status = A.overflow ? 1 : 0, count = 0, R16 = 0;''')
rep('''            const bool half16 = m < 32768 && n < 32768 && count < 32768;''','''            c2 = wall_clock64();
            const bool half16 = m < 32768 && n < 32768 && count < 32768;''')
rep('''                if (at + words > out_cap) status = 1; ''','''                c3 = wall_clock64();
                if (at + words > out_cap) status = 1; ''')
rep('''        po[0] = status; po[1] = count;''','''        { const unsigned long long c4 = wall_clock64(); uint32_t *g = gb_stamps + 5 * (slot & 131071u);
          g[0] = (uint32_t)(c1 - c0); g[1] = (uint32_t)(c2 - c1); g[2] = (uint32_t)(c3 - c2); g[3] = (uint32_t)(c4 - c3); g[4] = (uint32_t)c0; }
        po[0] = status; po[1] = count;''')
rep('int malcs_gbuild_pair_out_words() { return PAIR_OUT; }','''int malcs_gbuild_pair_out_words() { return PAIR_OUT; }
extern "C" int malcs_dev_gb_clocks(uint32_t *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gb_stamps), 5 * 131072 * 4); }''')
open(p,'w').write(s)
PY
mkdir -p ../../build_tools
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-value -c malcs_gbuild.hip -o /tmp/gbuild_clk.o
objs=$(ls build/*.o | grep -v 'build/malcs_gbuild.o')
hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_tools/libmalcs_gbclk.so $objs /tmp/gbuild_clk.o -ldl
echo "built build_tools/libmalcs_gbclk.so"
