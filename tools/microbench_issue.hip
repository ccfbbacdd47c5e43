// Tooling: issue cost of the instructions the generic overlap loop is made of, on the box's MI355X.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_issue.hip -o /tmp/mb && /tmp/mb
// Each kernel runs ITER trips of UNROLL independent instructions in every wave; 1, 2, 4 and 8 waves per SIMD.
// (the pair row counts a subtract + max as one; the LDS rows share one LDS among the CU's four SIMDs: divide by 4 for
// LDS cycles per instruction; the scalar rows are 64-byte loads, two in flight per wave)
// Printed: nanoseconds and (at the measured shader clock) cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4096
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum Op { ADD_U32, ADD_F64, MAX_F64, FMA_F64, SUBMAX_F64, LDS_B64_DISTINCT, LDS_B64_OFFSETS, LDS_B64_SAMEADDR, SMEM_X16_K, SMEM_X16_L2, SMEM_X1_HOT, SMEM_X2_HOT, SMEM_X4_HOT, SMEM_X8_HOT, SMEM_X16_HOT, LDS_OFFS_DEST_OVERLAPS_ADDR, LDS_OFFS_DEST_APART, NOPS };
static const char* NAMES[] = {"v_add_u32", "v_add_f64", "v_max_f64", "v_fma_f64", "v_add_f64+v_max_f64 (per pair)",
                              "ds_read_b64, 12 address VGPRs", "ds_read_b64, 4 address VGPRs x 3 immediate offsets",
                              "ds_read_b64, one address VGPR", "s_load_dwordx16, 4 KB per wave (scalar cache)",
                              "s_load_dwordx16, 256 KB per wave (L2)", "s_load_dword, one hot 1 KB block (kernarg-like)",
                              "s_load_dwordx2, hot block", "s_load_dwordx4, hot block", "s_load_dwordx8, hot block",
                              "s_load_dwordx16, hot block",
                              "ds_read_b64 x3 on one address VGPR, last destination overlaps it",
                              "ds_read_b64 x3 on one address VGPR, destinations apart"};

template <int OP>
__global__ void __launch_bounds__(1024) k(double* out, int seed, const char* cells, int span) {
    __shared__ double lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 0.5;
    __syncthreads();
    double a[12];
    uint32_t u[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) { a[j] = seed + j + lane; u[j] = seed * 8 + j * 8 * 64 + lane * 8; }
    const double s = seed * 1.5;
    uint32_t ad[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) ad[j] = (uint32_t)(uintptr_t)lds + (u[j] & 0x3FF8);
    for (int it = 0; it < ITER; ++it) {
        if (OP == ADD_U32) {
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[j]) : "s"(seed));
        } else if (OP == ADD_F64) {
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "s"(s));
        } else if (OP == MAX_F64) {
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[j]) : "s"(s));
        } else if (OP == FMA_F64) {
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[j]) : "s"(s));
        } else if (OP == SUBMAX_F64) {
            double t[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("v_add_f64 %0, %1, -%2" : "=v"(t[j]) : "v"(a[(j + 1) % 12]), "s"(s));
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[j]) : "v"(t[j]));
        } else if (OP == LDS_OFFS_DEST_OVERLAPS_ADDR || OP == LDS_OFFS_DEST_APART) {
            // does a destination that overlaps the address register of reads still in flight cost anything?
            // (fixed registers; 12 reads per trip as four blocks of three, one wait per trip)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (OP == LDS_OFFS_DEST_OVERLAPS_ADDR)
                    asm volatile("v_mov_b32 v40, %0\n ds_read_b64 v[42:43], v40\n ds_read_b64 v[44:45], v40 offset:512\n"
                                 "ds_read_b64 v[40:41], v40 offset:1024" :: "v"(ad[j]) : "v40", "v41", "v42", "v43", "v44", "v45");
                else
                    asm volatile("v_mov_b32 v40, %0\n ds_read_b64 v[42:43], v40\n ds_read_b64 v[44:45], v40 offset:512\n"
                                 "ds_read_b64 v[46:47], v40 offset:1024" :: "v"(ad[j]) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
        } else if (OP >= SMEM_X1_HOT && OP <= SMEM_X16_HOT) {
            // every wave of the chip reads the same 1 KB (what Params / ShapeRot reads of a transition kernel look like)
#pragma unroll
            for (int j = 0; j < 12; j += 2) {
                const char* q = cells + (((it * 12 + j) * 64) & 1023);
                if (OP == SMEM_X1_HOT) {
                    int x, y;
                    asm volatile("s_load_dword %0, %1, 0x0" : "=&s"(x) : "s"(q));
                    asm volatile("s_load_dword %0, %1, 0x20" : "=&s"(y) : "s"(q));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x), "+s"(y));
                    u[j] += x + y;
                } else if (OP == SMEM_X2_HOT) {
                    typedef int v2 __attribute__((ext_vector_type(2)));
                    v2 x, y;
                    asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=&s"(x) : "s"(q));
                    asm volatile("s_load_dwordx2 %0, %1, 0x20" : "=&s"(y) : "s"(q));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x), "+s"(y));
                    u[j] += x[1] + y[0];
                } else if (OP == SMEM_X4_HOT) {
                    typedef int v4 __attribute__((ext_vector_type(4)));
                    v4 x, y;
                    asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=&s"(x) : "s"(q));
                    asm volatile("s_load_dwordx4 %0, %1, 0x20" : "=&s"(y) : "s"(q));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x), "+s"(y));
                    u[j] += x[3] + y[1];
                } else if (OP == SMEM_X8_HOT) {
                    typedef int v8 __attribute__((ext_vector_type(8)));
                    v8 x, y;
                    asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(x) : "s"(q));
                    asm volatile("s_load_dwordx8 %0, %1, 0x20" : "=&s"(y) : "s"(q));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x), "+s"(y));
                    u[j] += x[3] + y[5];
                } else {
                    typedef int v16 __attribute__((ext_vector_type(16)));
                    v16 x, y;
                    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(x) : "s"(q));
                    asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=&s"(y) : "s"(q));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x), "+s"(y));
                    u[j] += x[3] + y[5];
                }
            }
        } else if (OP == SMEM_X16_K || OP == SMEM_X16_L2) {
            typedef int v16 __attribute__((ext_vector_type(16)));
            const int wv = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
            const char* p0 = cells + (size_t)(wv & 1023) * span;
#pragma unroll
            for (int j = 0; j < 12; j += 2) {
                v16 x, y;
                const char* q = p0 + ((it * 12 + j) * 64 & (span - 1));
                asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(x) : "s"(q));
                asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=&s"(y) : "s"(q));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x), "+s"(y));
                u[j] += x[3] + y[5];
            }
        } else {
            double t[12];
            if (OP == LDS_B64_DISTINCT) {
#pragma unroll
                for (int j = 0; j < 12; ++j) asm volatile("ds_read_b64 %0, %1" : "=v"(t[j]) : "v"(ad[j]));
            } else if (OP == LDS_B64_OFFSETS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("ds_read_b64 %0, %1" : "=v"(t[j]) : "v"(ad[j]));
                    asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(t[j + 4]) : "v"(ad[j]));
                    asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(t[j + 8]) : "v"(ad[j]));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 12; ++j) asm volatile("ds_read_b64 %0, %1" : "=v"(t[j]) : "v"(ad[0]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("" : "+v"(t[j]));
            if (it == ITER - 1) a[0] += t[0] + t[5] + t[11];
        }
    }
    double r = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) r += a[j] + u[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
static int run(double* out, const char* cells, double mhz, int cus) {
    for (int wps = 1; wps <= 8; wps *= 2) {
        // one workgroup of 4*wps waves per CU (a workgroup's waves are spread over the CU's four SIMDs)
        const int block = 64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps, grid = cus * (64 * 4 * wps / block);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(block), 0, 0, out, 3, cells, OP == SMEM_X16_K ? 4096 : 262144);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(block), 0, 0, out, 3, cells, OP == SMEM_X16_K ? 4096 : 262144);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per = OP == SUBMAX_F64 ? 12.0 : 12.0;          // instructions (pairs) per trip
        const double inst_per_simd = (double)ITER * per * wps;
        const double ns = ms * 1e6 / inst_per_simd;
        printf("  %-52s %d waves/SIMD: %7.3f ns = %6.2f cycles per wave-instruction per SIMD\n", NAMES[OP], wps, ns, ns * mhz * 1e-3);
    }
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const double mhz = p.clockRate / 1000.0;
    printf("%s, %d CUs, %.0f MHz (reported peak shader clock)\n", p.name, p.multiProcessorCount, mhz);
    double* out;
    CK(hipMalloc(&out, sizeof(double) * 1024 * 1024 * 4));
    const int cus = p.multiProcessorCount;
    char* cells;
    CK(hipMalloc(&cells, (size_t)1024 * 262144));
    CK(hipMemset(cells, 1, (size_t)1024 * 262144));
    if (run<ADD_U32>(out, cells, mhz, cus) || run<ADD_F64>(out, cells, mhz, cus) || run<MAX_F64>(out, cells, mhz, cus) ||
        run<FMA_F64>(out, cells, mhz, cus) || run<SUBMAX_F64>(out, cells, mhz, cus) || run<LDS_B64_DISTINCT>(out, cells, mhz, cus) ||
        run<LDS_B64_OFFSETS>(out, cells, mhz, cus) || run<LDS_B64_SAMEADDR>(out, cells, mhz, cus) ||
        run<SMEM_X16_K>(out, cells, mhz, cus) || run<SMEM_X16_L2>(out, cells, mhz, cus) ||
        run<SMEM_X1_HOT>(out, cells, mhz, cus) || run<SMEM_X2_HOT>(out, cells, mhz, cus) || run<SMEM_X4_HOT>(out, cells, mhz, cus) ||
        run<SMEM_X8_HOT>(out, cells, mhz, cus) || run<SMEM_X16_HOT>(out, cells, mhz, cus) ||
        run<LDS_OFFS_DEST_OVERLAPS_ADDR>(out, cells, mhz, cus) || run<LDS_OFFS_DEST_APART>(out, cells, mhz, cus)) return 1;
    return 0;
}
