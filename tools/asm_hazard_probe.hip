// What hipcc's hazard recogniser does and does not do around inline asm on gfx950 (tools/asm_hazards.py).
//   hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only -o probe.s tools/asm_hazard_probe.hip
// Each pair is the same dependency once with an ordinary consumer and once with the consumer inside asm.
#include <hip/hip_runtime.h>

// VALU writes a VGPR, v_readlane reads it (1 wait state): inserted for an ordinary and for an asm producer alike
__global__ void readlane_after_valu(const float *a, int *o) {
    float x = a[threadIdx.x];
    int r = (int)__builtin_floorf(x * 3.0f);
    o[threadIdx.x] = __builtin_amdgcn_readlane(r, 36) + r;
}
__global__ void readlane_after_asm_valu(const float *a, int *o) {
    float x = a[threadIdx.x];
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    o[threadIdx.x] = __builtin_amdgcn_readlane(r, 36);
}
// trans op writes a VGPR, a VALU instruction reads it (1 wait state): NOT inserted when the reader is asm
__global__ void valu_after_trans(const float *a, float *o) {
    float y = __builtin_amdgcn_rcpf(a[threadIdx.x]);
    o[threadIdx.x] = __builtin_floorf(y);
}
__global__ void asm_valu_after_trans(const float *a, int *o) {
    float y = __builtin_amdgcn_rcpf(a[threadIdx.x]);
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(y));
    o[threadIdx.x] = r;
}
// VALU writes an SGPR (v_readlane), a VALU instruction reads it (2 wait states): NOT inserted when the reader is asm
__global__ void valu_after_sgpr_write(const float *a, int *o) {
    float x = a[threadIdx.x];
    unsigned ry = __builtin_amdgcn_readlane(__float_as_int(x), 3);
    o[threadIdx.x] = (int)__builtin_amdgcn_fmed3f(x, __int_as_float(ry), 2.0f);
}
__global__ void asm_valu_after_sgpr_write(const float *a, int *o, int tile) {
    float x = a[threadIdx.x];
    unsigned ry = __builtin_amdgcn_readlane(__float_as_int(x), 3);
    unsigned row_addr;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row_addr) : "v"(threadIdx.x), "s"(ry), "v"(tile));
    o[threadIdx.x] = row_addr;
}
