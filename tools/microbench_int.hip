// Tooling: issue cost of the integer and cross-lane instructions the contour kernels are made of, on the box's MI355X.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_int.hip -o /tmp/mbi && /tmp/mbi
// Each kernel runs ITER trips of 12 independent instructions in every wave; 1, 2, 4 and 8 waves per SIMD.
// Printed: cycles per wave-instruction per SIMD at the reported shader clock (v_add_u32 is the 4-cycle yardstick).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4096
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum Op { ADD_U32, MUL_LO_U32, MUL_HI_U32, MUL_I24, MAD_I24, MAD_U64_U32, PERM_B32, BFE_U32, READLANE, DPP_ROW_SHR, BPERMUTE, FFBL, CNDMASK, LSHL_ADD, CNDMASK_SGPR, CMP_CNDMASK, BFI, MAX_DPP, ADD_SGPR_SRC, NOPS };
static const char* NAMES[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_i32_i24", "v_mad_i32_i24", "v_mad_u64_u32", "v_perm_b32",
                              "v_bfe_u32", "v_readlane_b32 (+ v_add with the SGPR)", "v_mov_b32 dpp row_shr:1", "ds_bpermute_b32", "v_ffbl_b32",
                              "v_cndmask_b32 (vcc, no writer in the loop)", "v_lshl_add_u32", "v_cndmask_b32_e64 (SGPR pair)", "v_cmp_lt_u32 + v_cndmask_b32 (pair)", "v_bfi_b32", "v_max_u32 dpp row_shr:1", "v_add_u32 with a fresh s_mov SGPR"};

template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t* out, int seed) {
    const int lane = threadIdx.x & 63;
    uint32_t u[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) u[j] = seed * 8 + j * 8 * 64 + lane * 8;
    unsigned long long w = (unsigned long long)seed * 77ull + lane;
    const unsigned long long w0 = __ballot((lane ^ seed) & 1);
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            if (OP == ADD_U32) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[j]) : "s"(seed));
            else if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(u[j]) : "s"(seed));
            else if (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(u[j]) : "s"(seed));
            else if (OP == MUL_I24) asm volatile("v_mul_i32_i24 %0, %1, %0" : "+v"(u[j]) : "s"(seed));
            else if (OP == MAD_I24) asm volatile("v_mad_i32_i24 %0, %1, %0, %0" : "+v"(u[j]) : "s"(seed));
            else if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w) : "v"(u[j]), "s"(seed) : "vcc");
            else if (OP == PERM_B32) asm volatile("v_perm_b32 %0, %1, %0, %0" : "+v"(u[j]) : "s"(seed));
            else if (OP == BFE_U32) asm volatile("v_bfe_u32 %0, %0, %1, 7" : "+v"(u[j]) : "s"(seed));
            else if (OP == READLANE) { int s; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(u[j])); asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[(j + 5) % 12]) : "s"(s)); }
            else if (OP == DPP_ROW_SHR) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[j]));
            else if (OP == BPERMUTE) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(u[j]) : "v"(u[(j + 1) % 12]));
            else if (OP == FFBL) asm volatile("v_ffbl_b32 %0, %0" : "+v"(u[j]));
            else if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(u[(j + 1) % 12]));
            else if (OP == CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) % 12]), "s"(w0));
            else if (OP == CMP_CNDMASK) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(u[(j + 1) % 12]) : "vcc");
            else if (OP == BFI) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) % 12]));
            else if (OP == MAX_DPP) asm volatile("s_nop 1\n v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[j]));
            else if (OP == ADD_SGPR_SRC) { int sg; asm volatile("s_mov_b32 %0, %1" : "=s"(sg) : "s"(seed)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[j]) : "s"(sg)); }
            else if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[j]) : "s"(seed));
        }
        if (OP == BPERMUTE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    uint32_t r = (uint32_t)w;
#pragma unroll
    for (int j = 0; j < 12; ++j) r += u[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
static int run(uint32_t* out, double mhz, int cus) {
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int block = 64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps, grid = cus * (64 * 4 * wps / block);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(block), 0, 0, out, 3);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(block), 0, 0, out, 3);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per = OP == READLANE ? 12.0 : 12.0;
        const double ns = ms * 1e6 / ((double)ITER * per * wps);
        printf("  %-44s %d waves/SIMD: %7.3f ns = %6.2f cycles per wave-instruction%s per SIMD\n", NAMES[OP], wps, ns, ns * mhz * 1e-3,
               OP == READLANE ? " pair" : "");
    }
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const double mhz = p.clockRate / 1000.0;
    printf("%s, %d CUs, %.0f MHz (reported peak shader clock)\n", p.name, p.multiProcessorCount, mhz);
    uint32_t* out;
    CK(hipMalloc(&out, sizeof(uint32_t) * 1024 * 1024 * 4));
    const int cus = p.multiProcessorCount;
    if (run<ADD_U32>(out, mhz, cus) || run<MUL_LO_U32>(out, mhz, cus) || run<MUL_HI_U32>(out, mhz, cus) || run<MUL_I24>(out, mhz, cus) ||
        run<MAD_I24>(out, mhz, cus) || run<MAD_U64_U32>(out, mhz, cus) || run<PERM_B32>(out, mhz, cus) || run<BFE_U32>(out, mhz, cus) ||
        run<READLANE>(out, mhz, cus) || run<DPP_ROW_SHR>(out, mhz, cus) || run<BPERMUTE>(out, mhz, cus) || run<FFBL>(out, mhz, cus) ||
        run<CNDMASK>(out, mhz, cus) || run<LSHL_ADD>(out, mhz, cus) || run<CNDMASK_SGPR>(out, mhz, cus) || run<CMP_CNDMASK>(out, mhz, cus) ||
        run<BFI>(out, mhz, cus) || run<MAX_DPP>(out, mhz, cus) || run<ADD_SGPR_SRC>(out, mhz, cus)) return 1;
    return 0;
}
