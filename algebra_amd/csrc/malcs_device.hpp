// malcs_device.hpp -- device helpers shared by the kernel families: the carry-chained limb step and
// the scalar-unit plumbing of the systolic lane array.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace malcs_dev {

// One limb of one column: U = V & Match (NP v_bitop3_b32, the match mask never materialises),
// S = V + U + carry (v_addc_co_u32 with the carry in an SGPR pair, so that it can be handed from limb
// to limb, from lane to lane and from step to step without touching VALU registers).
// The two v_bitop3 in front of every v_addc also provide the wait states gfx950 needs between a VALU
// write of an SGPR and a VALU read of it, which the compiler cannot see inside inline asm.
template <int NP, bool MASKED> struct limb;

template <> struct limb<2, false> {
    static __device__ __forceinline__ void step(uint32_t &U, uint32_t &S, uint64_t &cout, uint32_t V, uint64_t cin, uint32_t,
                                                const uint32_t *P, const uint32_t *x)
    {
        asm("v_bitop3_b32 %0, %3, %5, %6 bitop3:0x90\n\tv_bitop3_b32 %0, %0, %7, %8 bitop3:0x90\n\t"
            "v_addc_co_u32_e64 %1, %2, %3, %0, %4"
            : "=&v"(U), "=&v"(S), "=&s"(cout) : "v"(V), "s"(cin), "v"(P[0]), "v"(x[0]), "v"(P[1]), "v"(x[1]));
    }
};
template <> struct limb<2, true> {
    static __device__ __forceinline__ void step(uint32_t &U, uint32_t &S, uint64_t &cout, uint32_t V, uint64_t cin, uint32_t act,
                                                const uint32_t *P, const uint32_t *x)
    {
        asm("v_bitop3_b32 %0, %3, %6, %7 bitop3:0x90\n\tv_bitop3_b32 %0, %0, %8, %9 bitop3:0x90\n\t"
            "v_and_b32 %0, %0, %5\n\tv_addc_co_u32_e64 %1, %2, %3, %0, %4"
            : "=&v"(U), "=&v"(S), "=&s"(cout) : "v"(V), "s"(cin), "v"(act), "v"(P[0]), "v"(x[0]), "v"(P[1]), "v"(x[1]));
    }
};
// generic plane counts: the match mask is folded plane by plane in C (the compiler emits v_bitop3),
// only the carry-chained addition is asm; s_nop 1 supplies the wait states
template <int NP, bool MASKED> struct limb {
    static __device__ __forceinline__ void step(uint32_t &U, uint32_t &S, uint64_t &cout, uint32_t V, uint64_t cin, uint32_t act,
                                                const uint32_t *P, const uint32_t *x)
    {
        uint32_t u = V;
#pragma unroll
        for (int p = 0; p < NP; p++) u &= ~(P[p] ^ x[p]);
        if (MASKED) u &= act;
        U = u;
        asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(S), "=&s"(cout) : "v"(V), "v"(u), "s"(cin));
    }
};

// carry out of every lane -> carry into the next lane of the same pair (scalar unit, lane masks).
// Written as asm so that the value provably stays in SGPRs between the v_addc of one step and the next.
__device__ __forceinline__ uint64_t carry_to_next_lane(uint64_t cout, uint64_t segmask)
{
    uint64_t cin;
    asm("s_lshl_b64 %0, %1, 1\n\ts_and_b64 %0, %0, %2" : "=&s"(cin) : "s"(cout), "s"(segmask) : "scc");
    return cin;
}

// One column for one lane: chain the limbs, return the carry out of the lane's top limb.
template <int NL, int NP, bool MASKED>
__device__ __forceinline__ uint64_t column_step(uint32_t (&V)[NL], const uint32_t (&P)[NL][NP],
                                                const uint32_t (&x)[NP], uint32_t act, uint64_t cin)
{
    uint64_t c = cin;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        uint32_t U, S;
        uint64_t co;
        limb<NP, MASKED>::step(U, S, co, V[k], c, act, P[k], x);
        V[k] = S | (V[k] ^ U);
        c = co;
    }
    return c;
}


// S = a + b + carry (lane masks in SGPR pairs); s_nop supplies the wait states after a preceding v_addc
__device__ __forceinline__ uint32_t addc(uint32_t a, uint32_t b, uint64_t cin, uint64_t &cout)
{
    uint32_t s;
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(s), "=&s"(cout) : "v"(a), "v"(b), "s"(cin));
    return s;
}

// the window of row groups moves one lane down: so do the carries in flight
__device__ __forceinline__ uint64_t carry_window_down(uint64_t cin)
{
    uint64_t out;
    asm("s_lshr_b64 %0, %1, 1" : "=s"(out) : "s"(cin) : "scc");
    return out;
}

__device__ __forceinline__ uint64_t zero_mask()
{
    uint64_t z;                                    // an SGPR pair holding 0 (the asm needs a register, not a literal)
    asm("s_mov_b64 %0, 0" : "=s"(z));
    return z;
}

// One block of 32 symbols -> NP plane dwords: bit j of out[p] = bit p of the code of symbol j; symbols past
// `cnt` pack as code 0.  Returns the OR of all codes read (callers that validate the alphabet look at it).
template <int NP>
__device__ __forceinline__ uint32_t pack_block(const uint8_t *__restrict__ src, uint32_t cnt, uint32_t (&out)[NP])
{
#pragma unroll
    for (int p = 0; p < NP; p++) out[p] = 0;
    uint32_t seen = 0;
    for (uint32_t j = 0; j < cnt; j++) {
        const uint32_t c = src[j];
        seen |= c;
#pragma unroll
        for (int p = 0; p < NP; p++) out[p] |= ((c >> p) & 1u) << j;
    }
    return seen;
}

}  // namespace malcs_dev
