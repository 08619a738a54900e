// Probe: packed-f32 forms of complex arithmetic (v_pk_add / v_pk_mul / v_pk_fma with op_sel / neg modifiers) checked against
// scalar arithmetic -- the instruction forms behind csrc/vfi_fft.h's helpers.  hipcc --offload-arch=gfx950 -O3 pk_complex.hip -o bin/pk_complex
#include <hip/hip_runtime.h>
typedef float cpk __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cpk cmul(cpk A, cpk B) {
    cpk t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(A), "v"(B));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(A), "v"(B), "v"(t));
    return r;
}
__device__ __forceinline__ cpk cmulc(cpk A, cpk B) {   // a * conj(b)
    cpk t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(A), "v"(B));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(A), "v"(B), "v"(t));
    return r;
}
__device__ __forceinline__ cpk cmul_k(cpk A, float kr, float ki) {   // a * (kr + i ki), constants
    const cpk K = {kr, ki};
    cpk t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(A), "s"(K));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(A), "s"(K), "v"(t));
    return r;
}
__device__ __forceinline__ cpk add_rot(cpk T, cpk V) {   // t + (v.y, -v.x)
    cpk r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(T), "v"(V));
    return r;
}
__device__ __forceinline__ cpk sub_rot(cpk T, cpk V) {   // t - (v.y, -v.x)
    cpk r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(T), "v"(V));
    return r;
}
__global__ void k1(const cpk *x, const cpk *w, cpk *y) {
    int i = threadIdx.x;
    cpk a = x[i], b = x[i + 64], c = x[i + 128], d = x[i + 192];
    cpk s = a + b, t = a - b, u = c + d, v = c - d;
    y[i] = s + u; y[i + 64] = cmul(s - u, w[i]); y[i + 128] = cmulc(add_rot(t, v), w[i + 64]); y[i + 192] = cmul_k(sub_rot(t, v), 0.92387953f, -0.38268343f);
    y[i + 256] = (a * 0.86602540f) + b * -0.5f;
}
int main() {
    cpk *x, *w, *y; hipMallocManaged(&x, 2048); hipMallocManaged(&w, 1024); hipMallocManaged(&y, 4096);
    for (int i = 0; i < 256; ++i) x[i] = cpk{(float)(i % 7) - 3.f, (float)(i % 5) * 0.5f};
    for (int i = 0; i < 128; ++i) w[i] = cpk{0.6f, -0.8f};
    hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, 0, x, w, y); hipDeviceSynchronize();
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        cpk a = x[i], b = x[i + 64], c = x[i + 128], d = x[i + 192], W = w[i];
        cpk s = a + b, t = a - b, u = c + d, v = c - d;
        auto mul = [](cpk p, cpk q) { return cpk{p.x * q.x - p.y * q.y, p.x * q.y + p.y * q.x}; };
        cpk e1 = mul(s - u, W), r2 = t + cpk{v.y, -v.x}, e2 = mul(r2, cpk{W.x, -W.y}), r3 = t - cpk{v.y, -v.x}, e3 = mul(r3, cpk{0.92387953f, -0.38268343f});
        auto chk = [&](cpk g, cpk e) { if (fabsf(g.x - e.x) > 1e-5f || fabsf(g.y - e.y) > 1e-5f) ++bad; };
        chk(y[i + 64], e1); chk(y[i + 128], e2); chk(y[i + 192], e3);
    }
    printf("bad %d\n", bad);
}
