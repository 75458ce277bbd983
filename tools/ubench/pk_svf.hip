// pk_svf.hip -- GPU-box micro-benchmark and bit check: the state-variable filter's recurrence (Filter.zig:138-144) with its two pairs
// of products of the same b (cut * b, b * res) as ONE v_pk_mul_f32 each.  Forms of the same 32,768-sample walk per lane, one wave per CU:
//   plain     15 v_mul / v_add / v_sub per sample;
//   pk        the compiler's own v_pk_mul_f32 v[t], v[cut:res], v[b:..] op_sel_hi:[1,0] -- it puts an `s_nop 0` behind every one whose
//             result the next instruction reads (LLVM's dst-sel forwarding check reads src0's op_sel_hi bit of a VOP3P as a dst op_sel),
//             and a lone wave pays an issue slot for a nop as for anything else;
//   pk4_asm   four samples as one inline-asm block, 13 instructions a sample, no nop inside: b lives in either word of a register pair
//             (op_sel picks it), so the four b's handed on sit side by side for one 16-byte LDS write.
// Prints cycles per sample for each form and whether all leave the same bits in (l, b) and in a checksum of every sample's (l, b1).
// MEASURED (profiles/r06/ubench_pk_svf.txt): same bits, and NO gain -- 85.4 cycles a sample plain, 84.9 with 13 instructions: a
// v_pk_mul_f32 whose result the next instruction needs holds a lone wave for two issue slots (independent ones cost it 5.35 cycles,
// profiles/r01/ubench/valu_ops_mi355x.txt).  The compiler's form is slower where waves share a SIMD (nops and moves: NiceInstrument at
// 131,072 voices 144 against 122 us, profiles/r06/ab_pk_products.txt).  dsp.hip.h svf_core stays 15 plain instructions.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize pk_svf.hip -o pk_svf
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr float kDc = 3.814697265625e-6f;

struct Mid { float l, b1; };
template <int FORM>
__device__ __forceinline__ Mid step(float &l, float &b, float in, float cut, float res, v2f cr) {
    float cb, br;
    if (FORM == 0) { cb = cut * b; br = b * res; } else { const v2f t = cr * (v2f){b, b}; cb = t.x; br = t.y; }
    l += cb - kDc;
    b += cut * (in - br - l);
    if (FORM == 0) { cb = cut * b; br = b * res; } else { const v2f t = cr * (v2f){b, b}; cb = t.x; br = t.y; }
    l += cb;
    const float b1 = b;
    const float h = in - br - l;
    b += cut * h;
    return Mid{l, b1};
}

// one sample inside the block: B = the register b stands in (word W of its pair P), O = where b after :139 goes (word OW of pair OP), L0 -> L1
#define ZS_SAMPLE(P, SEL, LPREV, IN, O, OP, OSEL, LOUT)                                   \
    "v_pk_mul_f32 v[200:201], " P ", %[cr] " SEL "\n\t"                                   \
    "v_add_f32 v200, 0xb6800000, v200\n\t"                                                \
    "v_add_f32 v202, " LPREV ", v200\n\t"                                                 \
    "v_sub_f32 v201, " IN ", v201\n\t"                                                    \
    "v_sub_f32 v201, v201, v202\n\t"                                                      \
    "v_mul_f32 v201, %[cut], v201\n\t"                                                    \
    "v_add_f32 " O ", v204, v201\n\t"                                                     \
    "v_pk_mul_f32 v[200:201], " OP ", %[cr] " OSEL "\n\t"                                 \
    "v_add_f32 " LOUT ", v202, v200\n\t"                                                  \
    "v_sub_f32 v201, " IN ", v201\n\t"                                                    \
    "v_sub_f32 v201, v201, " LOUT "\n\t"                                                  \
    "v_mul_f32 v201, %[cut], v201\n\t"                                                    \
    "v_add_f32 v204, " O ", v201\n\t"
#define ZS_LO "op_sel_hi:[0,1]"
#define ZS_HI "op_sel:[1,0] op_sel_hi:[1,1]"

__device__ __forceinline__ void step4_asm(float &l, float &b, v4f x, float cut, v2f cr, v4f &lq, v4f &bq) {
    v2f t; float lt;
    asm(ZS_SAMPLE("v[204:205]", ZS_LO, "%[l]", "%[x0]", "v216", "v[216:217]", ZS_LO, "v220")
        ZS_SAMPLE("v[204:205]", ZS_LO, "v220", "%[x1]", "v217", "v[216:217]", ZS_HI, "v221")
        ZS_SAMPLE("v[204:205]", ZS_LO, "v221", "%[x2]", "v218", "v[218:219]", ZS_LO, "v222")
        ZS_SAMPLE("v[204:205]", ZS_LO, "v222", "%[x3]", "v219", "v[218:219]", ZS_HI, "v223")
        : "=&{v[216:219]}"(bq), "=&{v[220:223]}"(lq), "=&{v[200:201]}"(t), "=&{v202}"(lt), "+{v204}"(b)
        : [cr] "v"(cr), [cut] "v"(cut), [l] "v"(l), [x0] "v"(x.x), [x1] "v"(x.y), [x2] "v"(x.z), [x3] "v"(x.w)
        : "v205");
    l = lq.w;
}
__device__ __forceinline__ unsigned mix(unsigned a, float x) { return (a ^ __float_as_uint(x)) * 0x9E3779B1u; }

template <int FORM>
__global__ void __launch_bounds__(64) k_walk(unsigned *out, unsigned long long *cyc, const float *in, const float *cutp, const float *resp, unsigned n, bool sum) {
    const unsigned g = blockIdx.x * 64 + threadIdx.x;
    const float cut = cutp[g], res = resp[g];
    const v2f cr = {cut, res};
    float l = 0.0f, b = 0.0f;
    unsigned acc = 0;
    const float scale = 1.0f + 0.001f * (float)(threadIdx.x & 7);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (unsigned i = 0; i < n; i += 8) {
        v4f x[2];
#pragma unroll
        for (int k = 0; k < 2; k++) x[k] = *(const v4f *)&in[(i + 4 * k) & 1023u] * scale;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            v4f lq, bq;
            if (FORM == 2) step4_asm(l, b, x[k], cut, cr, lq, bq);
            else {
                const Mid m0 = step<FORM>(l, b, x[k].x, cut, res, cr), m1 = step<FORM>(l, b, x[k].y, cut, res, cr);
                const Mid m2 = step<FORM>(l, b, x[k].z, cut, res, cr), m3 = step<FORM>(l, b, x[k].w, cut, res, cr);
                lq = (v4f){m0.l, m1.l, m2.l, m3.l}; bq = (v4f){m0.b1, m1.b1, m2.b1, m3.b1};
            }
            if (sum) acc = mix(mix(mix(mix(mix(mix(mix(mix(acc, lq.x), bq.x), lq.y), bq.y), lq.z), bq.z), lq.w), bq.w);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[g * 3] = __float_as_uint(l); out[g * 3 + 1] = __float_as_uint(b); out[g * 3 + 2] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const unsigned B = 256, N = 32768;
    float *in, *cut, *res; unsigned *out[3]; unsigned long long *cyc;
    CK(hipMalloc(&in, 1024 * 4)); CK(hipMalloc(&cut, B * 64 * 4)); CK(hipMalloc(&res, B * 64 * 4)); CK(hipMalloc(&cyc, B * 8));
    for (auto &o : out) CK(hipMalloc(&o, B * 64 * 3 * 4));
    float hin[1024], *hc = (float *)malloc(B * 64 * 4), *hr = (float *)malloc(B * 64 * 4);
    srand(7);
    for (auto &v : hin) v = (rand() / (float)RAND_MAX) * 2.0f - 1.0f;
    for (unsigned i = 0; i < B * 64; i++) {
        hc[i] = (rand() / (float)RAND_MAX) * ((i & 15) == 0 ? 1e-30f : (i & 15) == 1 ? 1.0f : 0.6f);   // tiny cutoffs too: denormal products
        hr[i] = (i & 31) == 2 ? 0.0f : rand() / (float)RAND_MAX;
    }
    hin[17] = 1e-39f; hin[300] = -0.0f; hin[301] = 0.0f;               // a denormal input, zeros of both signs
    CK(hipMemcpy(in, hin, sizeof hin, hipMemcpyHostToDevice)); CK(hipMemcpy(cut, hc, B * 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(res, hr, B * 64 * 4, hipMemcpyHostToDevice));
    const char *names[3] = {"plain", "pk (compiler, nops)", "pk4_asm (no nop)"};
    unsigned *h[3];
    for (int sum = 1; sum >= 0; sum--) {
        for (int f = 0; f < 3; f++) {
            for (int rep = 0; rep < 2; rep++) {
                if (f == 0) hipLaunchKernelGGL(k_walk<0>, dim3(B), dim3(64), 0, 0, out[f], cyc, in, cut, res, N, sum != 0);
                if (f == 1) hipLaunchKernelGGL(k_walk<1>, dim3(B), dim3(64), 0, 0, out[f], cyc, in, cut, res, N, sum != 0);
                if (f == 2) hipLaunchKernelGGL(k_walk<2>, dim3(B), dim3(64), 0, 0, out[f], cyc, in, cut, res, N, sum != 0);
                CK(hipDeviceSynchronize());
            }
            unsigned long long hcyc[B]; CK(hipMemcpy(hcyc, cyc, sizeof hcyc, hipMemcpyDeviceToHost));
            double avg = 0; for (unsigned i = 0; i < B; i++) avg += (double)hcyc[i]; avg /= B;
            printf("%-22s %s: %.2f cycles per sample (__builtin_readcyclecounter)\n", names[f], sum ? "with checksum" : "chain alone  ", avg / N);
            if (sum) { h[f] = (unsigned *)malloc(B * 64 * 3 * 4); CK(hipMemcpy(h[f], out[f], B * 64 * 3 * 4, hipMemcpyDeviceToHost)); }
        }
    }
    for (int f = 1; f < 3; f++) {
        unsigned bad = 0;
        for (unsigned i = 0; i < B * 64 * 3; i++) bad += h[f][i] != h[0][i];
        printf("%s against plain: %u of %u words differ (final l, final b, checksum of every sample's l and b1 over %u samples x %u lanes)\n", names[f], bad, B * 64 * 3, N, B * 64);
    }
    return 0;
}
