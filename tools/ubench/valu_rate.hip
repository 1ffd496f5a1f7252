// valu_rate.hip -- issue-rate micro-benchmark for the integer VALU instructions the ORB / Hamming kernels are built from.
// Prints wave-instructions per cycle per SIMD assuming the clock reported by the device (and the raw Ginstr/s).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int kIters = 2048;

#define BODY8(INS)                                                                     \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)               \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(b), "v"(c));

#define KERNEL(NAME, INS)                                                              \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {        \
        uint32_t a[8];                                                                 \
        for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i;               \
        uint32_t b = seed ^ 0x00330044u, c = seed + 0x00110022u;                       \
        for (int it = 0; it < kIters; ++it) { BODY8(INS) BODY8(INS) }                  \
        uint32_t r = 0;                                                                \
        for (int i = 0; i < 8; ++i) r ^= a[i];                                         \
        if (r == 0x12345678u) out[threadIdx.x] = r;                                    \
    }

#define I_PK_MAX_I16(n) "v_pk_max_i16 %" #n ", %" #n ", %8\n"
#define I_PK_MIN_U16(n) "v_pk_min_u16 %" #n ", %" #n ", %8\n"
#define I_PK_MAX_F16(n) "v_pk_max_f16 %" #n ", %" #n ", %8\n"
#define I_PK_MAX3_F16(n) "v_pk_maximum3_f16 %" #n ", %" #n ", %8, %9\n"
#define I_PK_ADD_U16(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define I_MAX_I32(n) "v_max_i32 %" #n ", %" #n ", %8\n"
#define I_MAX3_I32(n) "v_max3_i32 %" #n ", %" #n ", %8, %9\n"
#define I_MAX_I16(n) "v_max_i16 %" #n ", %" #n ", %8\n"
#define I_MAX3_I16(n) "v_max3_i16 %" #n ", %" #n ", %8, %9\n"
#define I_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_BCNT(n) "v_bcnt_u32_b32 %" #n ", %8, %" #n "\n"
#define I_ADD_U32(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_ADD3_U32(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_FMA_F32(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PK_FMA_F32(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_AND_OR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define I_SAD_U8(n) "v_sad_u8 %" #n ", %" #n ", %8, %9\n"
#define I_MAD_U32_U24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define I_LSHL_OR(n) "v_lshl_or_b32 %" #n ", %" #n ", 3, %9\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_MAX_U16_SDWA(n) "v_max_u16_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"

// rotating sources: every instruction reads three (two) DIFFERENT accumulator registers, as real code does
#define BODY8R3(OP)                                                                    \
    asm volatile(OP " %0, %1, %2, %5\n" OP " %1, %2, %3, %6\n" OP " %2, %3, %4, %7\n" OP " %3, %4, %5, %0\n"      \
                 OP " %4, %5, %6, %1\n" OP " %5, %6, %7, %2\n" OP " %6, %7, %0, %3\n" OP " %7, %0, %1, %4\n"      \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
#define BODY8R2(OP)                                                                    \
    asm volatile(OP " %0, %1, %2\n" OP " %1, %2, %3\n" OP " %2, %3, %4\n" OP " %3, %4, %5\n"      \
                 OP " %4, %5, %6\n" OP " %5, %6, %7\n" OP " %6, %7, %0\n" OP " %7, %0, %1\n"      \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
#define BODY8R1(OP)                                                                    \
    asm volatile(OP " %0, %1\n" OP " %1, %2\n" OP " %2, %3\n" OP " %3, %4\n" OP " %4, %5\n" OP " %5, %6\n" OP " %6, %7\n" OP " %7, %0\n" \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
#define BODY8R3D(OP)                                                                   \
    asm volatile(OP " %0, %1, %2, %3\n" OP " %1, %2, %3, %0\n" OP " %2, %3, %0, %1\n" OP " %3, %0, %1, %2\n"      \
                 OP " %0, %1, %2, %3\n" OP " %1, %2, %3, %0\n" OP " %2, %3, %0, %1\n" OP " %3, %0, %1, %2\n"      \
                 : "+v"(*reinterpret_cast<double*>(&a[0])), "+v"(*reinterpret_cast<double*>(&a[2])), "+v"(*reinterpret_cast<double*>(&a[4])), \
                   "+v"(*reinterpret_cast<double*>(&a[6])));
#define KERNELR(NAME, BODY, OP)                                                        \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {        \
        uint32_t a[8];                                                                 \
        for (int i = 0; i < 8; ++i) a[i] = (seed * (threadIdx.x + 1) + i) & 0x00ff00ffu; \
        for (int it = 0; it < kIters; ++it) { BODY(OP) BODY(OP) }                      \
        uint32_t r = 0;                                                                \
        for (int i = 0; i < 8; ++i) r ^= a[i];                                         \
        if (r == 0x12345678u) out[threadIdx.x] = r;                                    \
    }
KERNELR(r_pk_max3_f16, BODY8R3, "v_pk_maximum3_f16")
KERNELR(r_pk_max_f16, BODY8R2, "v_pk_max_f16")
KERNELR(r_pk_max_i16, BODY8R2, "v_pk_max_i16")
KERNELR(r_perm, BODY8R3, "v_perm_b32")
KERNELR(r_max3_i32, BODY8R3, "v_max3_i32")
KERNELR(r_xor, BODY8R2, "v_xor_b32")
KERNELR(r_add_u32, BODY8R2, "v_add_u32")
KERNELR(r_max_i16, BODY8R2, "v_max_i16")
KERNELR(r_min_u16, BODY8R2, "v_min_u16")
KERNELR(r_and_or, BODY8R3, "v_and_or_b32")
KERNELR(r_fma_f32, BODY8R3, "v_fma_f32")
// round 2: the rest of the VOP2 / VOP3 integer vocabulary of the kernels (which forms issue at the fast rate?)
KERNELR(r_and, BODY8R2, "v_and_b32")
KERNELR(r_or, BODY8R2, "v_or_b32")
KERNELR(r_sub_u32, BODY8R2, "v_sub_u32")
KERNELR(r_lshlrev, BODY8R2, "v_lshlrev_b32")
KERNELR(r_lshrrev, BODY8R2, "v_lshrrev_b32")
KERNELR(r_max_u32, BODY8R2, "v_max_u32")
KERNELR(r_min_i32, BODY8R2, "v_min_i32")
KERNELR(r_mul_u32_u24, BODY8R2, "v_mul_u32_u24")
KERNELR(r_mul_hi_u32_u24, BODY8R2, "v_mul_hi_u32_u24")
KERNELR(r_mul_lo_u16, BODY8R2, "v_mul_lo_u16")
KERNELR(r_add_u16, BODY8R2, "v_add_u16")
KERNELR(r_sub_u16, BODY8R2, "v_sub_u16")
KERNELR(r_lshlrev_b16, BODY8R2, "v_lshlrev_b16")
KERNELR(r_ashrrev_i16, BODY8R2, "v_ashrrev_i16")
KERNELR(r_max_u16, BODY8R2, "v_max_u16")
KERNELR(r_bfe_u32, BODY8R3, "v_bfe_u32")
KERNELR(r_alignbit, BODY8R3, "v_alignbit_b32")
KERNELR(r_alignbyte, BODY8R3, "v_alignbyte_b32")
KERNELR(r_mad_u16, BODY8R3, "v_mad_u16")
KERNELR(r_mad_i32_i24, BODY8R3, "v_mad_i32_i24")
KERNELR(r_lshl_add, BODY8R3, "v_lshl_add_u32")
KERNELR(r_or3, BODY8R3, "v_or3_b32")
KERNELR(r_dot4_u32_u8, BODY8R3, "v_dot4_u32_u8")
KERNELR(r_dot2_u32_u16, BODY8R3, "v_dot2_u32_u16")
KERNELR(r_bfi, BODY8R3, "v_bfi_b32")
KERNELR(r_min3_u16, BODY8R3, "v_min3_u16")
KERNELR(r_med3_i32, BODY8R3, "v_med3_i32")
KERNELR(r_cndmask, BODY8R2, "v_cndmask_b32")
KERNELR(r_pk_sub_i16, BODY8R2, "v_pk_sub_i16")
KERNELR(r_pk_lshrrev_b16, BODY8R2, "v_pk_lshrrev_b16")
// round 3: candidates for the pyramid's vertical pass and the describe kernel
KERNELR(r_mad_u32_u16, BODY8R3, "v_mad_u32_u16")
KERNELR(r_mad_u32_u24, BODY8R3, "v_mad_u32_u24")
KERNELR(r_add3_u32, BODY8R3, "v_add3_u32")
KERNELR(r_lshl_or, BODY8R3, "v_lshl_or_b32")
KERNELR(r_cvt_f32_i32, BODY8R1, "v_cvt_f32_i32")
KERNELR(r_mul_f32, BODY8R2, "v_mul_f32")
KERNELR(r_add_f32, BODY8R2, "v_add_f32")
KERNELR(r_fma_f64, BODY8R3D, "v_fma_f64")

KERNEL(k_pk_max_i16, I_PK_MAX_I16)
KERNEL(k_pk_min_u16, I_PK_MIN_U16)
KERNEL(k_pk_max_f16, I_PK_MAX_F16)
KERNEL(k_pk_max3_f16, I_PK_MAX3_F16)
KERNEL(k_pk_add_u16, I_PK_ADD_U16)
KERNEL(k_perm, I_PERM)
KERNEL(k_max_i32, I_MAX_I32)
KERNEL(k_max3_i32, I_MAX3_I32)
KERNEL(k_max_i16, I_MAX_I16)
KERNEL(k_max3_i16, I_MAX3_I16)
KERNEL(k_xor, I_XOR)
KERNEL(k_bcnt, I_BCNT)
KERNEL(k_add_u32, I_ADD_U32)
KERNEL(k_add3_u32, I_ADD3_U32)
KERNEL(k_fma_f32, I_FMA_F32)
KERNEL(k_and_or, I_AND_OR)
KERNEL(k_sad_u8, I_SAD_U8)
KERNEL(k_mad_u32_u24, I_MAD_U32_U24)
KERNEL(k_lshl_or, I_LSHL_OR)
KERNEL(k_mov, I_MOV)
KERNEL(k_max_u16_sdwa, I_MAX_U16_SDWA)

// Do the packed f16 min/max instructions order u16 bit patterns 0..255 (f16 denormals) like integers, and return them unflushed?
__global__ void k_f16_denorm_check(uint32_t* bad) {
    const uint32_t a = threadIdx.x, b = blockIdx.x;   // 256 x 256
    uint32_t n = 0;
    for (uint32_t c = 0; c < 256; c += 5) {
        const uint32_t pa = a | (b << 16), pb = b | (c << 16), pc = c | (a << 16);
        uint32_t mx3, mn3, mx2, mn2;
        asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(mx3) : "v"(pa), "v"(pb), "v"(pc));
        asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(mn3) : "v"(pa), "v"(pb), "v"(pc));
        asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(mx2) : "v"(pa), "v"(pb));
        asm volatile("v_pk_min_f16 %0, %1, %2" : "=v"(mn2) : "v"(pa), "v"(pb));
        auto mx = [](uint32_t x, uint32_t y) { return x > y ? x : y; };
        auto mn = [](uint32_t x, uint32_t y) { return x < y ? x : y; };
        const uint32_t e_mx3 = mx(mx(a, b), c) | (mx(mx(b, c), a) << 16), e_mn3 = mn(mn(a, b), c) | (mn(mn(b, c), a) << 16);
        const uint32_t e_mx2 = mx(a, b) | (mx(b, c) << 16), e_mn2 = mn(a, b) | (mn(b, c) << 16);
        n += (mx3 != e_mx3) + (mn3 != e_mn3);
        n += ((mx2 != e_mx2) + (mn2 != e_mn2)) << 16;
    }
    if (n) atomicAdd(bad, n);
}

// ---- round 5: MIXED streams. The tables above time one opcode back to back; k_fast_cells' pre-test issues a mix of fast-class
// VOP2 ops (v_sub_u32 / v_add_u32 / v_or_b32: 2.3-2.5 cycles alone) and VOP3 ops (v_alignbyte_b32, v_bitop3_b32: 4.2-4.4 alone) with real
// dependences. Does a mixed stream issue at the weighted mean of the single-opcode rates (~3.0), or at 4 cycles per instruction whatever the opcode
// (what SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU on the kernel suggested)? One body = swar_diameter_test's 32 instructions for one 4-pixel group:
// 5 v_alignbyte, 2 + 16 v_sub / v_add, 2 v_or, 6 v_bitop3, 1 v_and. Run at 4, 7 and 8 waves per SIMD (the kernel runs at 7).
#define MIX_PRETEST_BODY                                                                                   \
    asm volatile(                                                                                          \
        "v_alignbyte_b32 %8, %1, %0, 2\n v_alignbyte_b32 %9, %2, %1, 2\n v_alignbyte_b32 %10, %3, %2, 2\n" \
        "v_alignbyte_b32 %11, %4, %3, 1\n v_alignbyte_b32 %12, %5, %4, 3\n"                                \
        "v_sub_u32 %13, %8, %6\n v_add_u32 %14, %8, %6\n"                                                  \
        "v_sub_u32 %0, %9, %13\n v_sub_u32 %1, %10, %13\n v_sub_u32 %2, %11, %13\n v_sub_u32 %3, %12, %13\n" \
        "v_sub_u32 %4, %5, %13\n v_sub_u32 %5, %7, %13\n v_sub_u32 %9, %6, %13\n v_sub_u32 %10, %7, %13\n"   \
        "v_or_b32 %0, %0, %1\n"                                                                            \
        "v_bitop3_b32 %0, %0, %2, %3 bitop3:0xe0\n v_bitop3_b32 %0, %0, %4, %5 bitop3:0xe0\n v_bitop3_b32 %0, %0, %9, %10 bitop3:0xe0\n" \
        "v_sub_u32 %1, %14, %11\n v_sub_u32 %2, %14, %12\n v_sub_u32 %3, %14, %8\n v_sub_u32 %4, %14, %7\n"  \
        "v_sub_u32 %5, %14, %6\n v_sub_u32 %9, %14, %0\n v_sub_u32 %10, %14, %13\n v_sub_u32 %11, %14, %12\n" \
        "v_or_b32 %1, %1, %2\n"                                                                            \
        "v_bitop3_b32 %1, %1, %3, %4 bitop3:0xe0\n v_bitop3_b32 %1, %1, %5, %9 bitop3:0xe0\n v_bitop3_b32 %1, %1, %10, %11 bitop3:0xe0\n" \
        "v_and_b32 %2, %0, %1\n"                                                                           \
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]),     \
          "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]));
// the exact scorer's mix: 16-bit VOP2 min / max (2.3 alone) with a v_perm_b32 / v_bfe_u32 (4.2 alone) every fourth instruction
#define MIX_SCORE_BODY                                                                                     \
    asm volatile(                                                                                          \
        "v_min_u16 %8, %0, %1\n v_min_u16 %9, %2, %3\n v_min_u16 %10, %4, %5\n v_bfe_u32 %11, %6, 8, 8\n"   \
        "v_max_u16 %0, %8, %9\n v_min_u16 %1, %9, %10\n v_max_u16 %2, %10, %11\n v_perm_b32 %3, %7, %6, %5\n" \
        "v_min_u16 %4, %0, %1\n v_max_u16 %5, %1, %2\n v_min_u16 %6, %2, %3\n v_bfe_u32 %7, %3, 16, 8\n"    \
        "v_max_u16 %8, %4, %5\n v_min_u16 %9, %5, %6\n v_max_u16 %10, %6, %7\n v_perm_b32 %11, %4, %7, %0\n" \
        "v_min_u16 %0, %8, %9\n v_min_u16 %1, %9, %10\n v_min_u16 %2, %10, %11\n v_bfe_u32 %3, %11, 8, 8\n"  \
        "v_max_u16 %4, %0, %1\n v_min_u16 %5, %1, %2\n v_max_u16 %6, %2, %3\n v_perm_b32 %7, %3, %2, %1\n"   \
        "v_min_u16 %8, %4, %5\n v_max_u16 %9, %5, %6\n v_min_u16 %10, %6, %7\n v_bfe_u32 %11, %7, 16, 8\n"  \
        "v_max_u16 %0, %8, %9\n v_min_u16 %1, %9, %10\n v_max_u16 %2, %10, %11\n v_perm_b32 %3, %8, %11, %4\n" \
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]),     \
          "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]));
// pure fast-class stream with the same dependence shape (reference for the two above)
#define MIX_VOP2_BODY                                                                                      \
    asm volatile(                                                                                          \
        "v_sub_u32 %8, %0, %1\n v_add_u32 %9, %2, %3\n v_or_b32 %10, %4, %5\n v_and_b32 %11, %6, %7\n"     \
        "v_sub_u32 %0, %8, %9\n v_add_u32 %1, %9, %10\n v_or_b32 %2, %10, %11\n v_and_b32 %3, %7, %6\n"     \
        "v_sub_u32 %4, %0, %1\n v_add_u32 %5, %1, %2\n v_or_b32 %6, %2, %3\n v_xor_b32 %7, %3, %4\n"        \
        "v_sub_u32 %8, %4, %5\n v_add_u32 %9, %5, %6\n v_or_b32 %10, %6, %7\n v_and_b32 %11, %4, %7\n"      \
        "v_sub_u32 %0, %8, %9\n v_add_u32 %1, %9, %10\n v_or_b32 %2, %10, %11\n v_xor_b32 %3, %11, %8\n"    \
        "v_sub_u32 %4, %0, %1\n v_add_u32 %5, %1, %2\n v_or_b32 %6, %2, %3\n v_and_b32 %7, %3, %2\n"        \
        "v_sub_u32 %8, %4, %5\n v_add_u32 %9, %5, %6\n v_or_b32 %10, %6, %7\n v_xor_b32 %11, %7, %4\n"      \
        "v_sub_u32 %0, %8, %9\n v_add_u32 %1, %9, %10\n v_or_b32 %2, %10, %11\n v_and_b32 %3, %8, %11\n"    \
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]),     \
          "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]));
#define KERNELMIX(NAME, BODY)                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {        \
        uint32_t a[8], t[7];                                                           \
        for (int i = 0; i < 8; ++i) a[i] = (seed * (threadIdx.x + 1) + i) | 0x80808080u; \
        for (int i = 0; i < 7; ++i) t[i] = seed + i;                                   \
        for (int it = 0; it < kIters / 2; ++it) { BODY }                               \
        uint32_t r = 0;                                                                \
        for (int i = 0; i < 8; ++i) r ^= a[i];                                         \
        for (int i = 0; i < 7; ++i) r ^= t[i];                                         \
        if (r == 0x12345678u) out[threadIdx.x] = r;                                    \
    }
KERNELMIX(mix_pretest, MIX_PRETEST_BODY)
KERNELMIX(mix_score, MIX_SCORE_BODY)
KERNELMIX(mix_vop2, MIX_VOP2_BODY)

template <typename K>
int run_mix(const char* name, K kern, uint32_t* d_out, int cus, double clock_ghz, int wg_per_cu) {
    dim3 grid(cus * wg_per_cu), block(256);   // wg_per_cu workgroups of 4 waves per CU = wg_per_cu waves per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, grid, block, 0, 0, d_out, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, block, 0, 0, d_out, 12345u + i);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)cus * wg_per_cu * 4 * (kIters / 2) * 32.0 * reps;   // 32 instructions per body
    const double per_simd_cycle = instr / (ms * 1e-3) / (cus * 4.0) / (clock_ghz * 1e9);
    printf("%-12s %d waves/SIMD  %8.3f ms  %.3f wave-instr/cycle/SIMD (@%.2f GHz) => %.2f cycles/instr\n", name, wg_per_cu, ms / reps, per_simd_cycle,
           clock_ghz, 1.0 / per_simd_cycle);
    return 0;
}

template <typename K>
int run(const char* name, K kern, uint32_t* d_out, int cus, double clock_ghz) {
    const int wg_per_cu = 8;   // 8 x 4 waves = 32 waves / CU = 8 waves / SIMD
    dim3 grid(cus * wg_per_cu), block(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, grid, block, 0, 0, d_out, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, block, 0, 0, d_out, 12345u + i);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)cus * wg_per_cu * 4;
    const double instr = waves * kIters * 16.0 * reps;
    const double per_s = instr / (ms * 1e-3);
    const double per_simd_cycle = per_s / (cus * 4.0) / (clock_ghz * 1e9);
    printf("%-18s %8.3f ms  %8.2f Gwave-instr/s  %.3f wave-instr/cycle/SIMD (@%.2f GHz) => %.2f cycles/instr\n", name, ms / reps, per_s / 1e9,
           per_simd_cycle, clock_ghz, 1.0 / per_simd_cycle);
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    printf("device %s, %d CUs, clockRate %.3f GHz\n", prop.gcnArchName, cus, ghz);
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, 4096));
    if (argc > 1 && argv[1][0] == 'm') {   // `valu_rate mix`: the mixed streams only
        for (int w : {1, 2, 4, 7, 8}) {
            if (run_mix("mix_vop2", mix_vop2, d_out, cus, ghz, w)) return 1;
            if (run_mix("mix_pretest", mix_pretest, d_out, cus, ghz, w)) return 1;
            if (run_mix("mix_score", mix_score, d_out, cus, ghz, w)) return 1;
        }
        return 0;
    }
    {
        uint32_t* d_bad;
        CHECK(hipMalloc(&d_bad, 4));
        CHECK(hipMemset(d_bad, 0, 4));
        hipLaunchKernelGGL(k_f16_denorm_check, dim3(256), dim3(256), 0, 0, d_bad);
        uint32_t bad = 0;
        CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        printf("f16 min/max on u8-in-u16 bit patterns: 3-input mismatches %u, 2-input mismatches %u (0 = usable as integer min/max)\n",
               bad & 0xFFFFu, bad >> 16);
    }
#define RUN(k) if (run(#k, k, d_out, cus, ghz)) return 1;
    RUN(k_fma_f32)
    RUN(k_mov)
    RUN(k_add_u32)
    RUN(k_add3_u32)
    RUN(k_xor)
    RUN(k_bcnt)
    RUN(k_and_or)
    RUN(k_lshl_or)
    RUN(k_perm)
    RUN(k_max_i32)
    RUN(k_max3_i32)
    RUN(k_max_i16)
    RUN(k_max3_i16)
    RUN(k_max_u16_sdwa)
    RUN(k_pk_max_i16)
    RUN(k_pk_min_u16)
    RUN(k_pk_add_u16)
    RUN(k_pk_max_f16)
    RUN(k_pk_max3_f16)
    RUN(k_sad_u8)
    RUN(r_pk_max3_f16)
    RUN(r_pk_max_f16)
    RUN(r_pk_max_i16)
    RUN(r_perm)
    RUN(r_max3_i32)
    RUN(r_and_or)
    RUN(r_xor)
    RUN(r_add_u32)
    RUN(r_max_i16)
    RUN(r_min_u16)
    RUN(r_fma_f32)
    RUN(k_mad_u32_u24)
    RUN(r_and) RUN(r_or) RUN(r_sub_u32) RUN(r_lshlrev) RUN(r_lshrrev) RUN(r_max_u32) RUN(r_min_i32) RUN(r_mul_u32_u24) RUN(r_mul_hi_u32_u24)
    RUN(r_mul_lo_u16) RUN(r_add_u16) RUN(r_sub_u16) RUN(r_lshlrev_b16) RUN(r_ashrrev_i16) RUN(r_max_u16) RUN(r_bfe_u32) RUN(r_alignbit)
    RUN(r_alignbyte) RUN(r_mad_u16) RUN(r_mad_i32_i24) RUN(r_lshl_add) RUN(r_or3) RUN(r_dot4_u32_u8) RUN(r_dot2_u32_u16) RUN(r_bfi)
    RUN(r_min3_u16) RUN(r_med3_i32) RUN(r_cndmask) RUN(r_pk_sub_i16) RUN(r_pk_lshrrev_b16)
    RUN(r_mad_u32_u16) RUN(r_mad_u32_u24) RUN(r_add3_u32) RUN(r_lshl_or) RUN(r_cvt_f32_i32) RUN(r_mul_f32) RUN(r_add_f32) RUN(r_fma_f64)
    return 0;
}
