// Does the range check of a raw buffer load (stride 0) on gfx950 cover the SCALAR offset?  A descriptor of 16 bytes over a 4 KB array of 7.0f:
// loads at (voffset, soffset, inst offset) inside and outside the 16 bytes.  A clipped load returns 0.
// build + run: hipcc -O3 --offload-arch=gfx950 tools/micro/buffer_soffset_check.hip -o /tmp/bsc && /tmp/bsc
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *p, float *out, int so)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 16, 0x00020000);
    out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0));            // inside
    out[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 64, 0, 0));           // voffset outside
    out[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 0, so, 0));           // soffset (SGPR) outside
    out[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 64, 0));           // constant: immediate / scalar as the compiler likes
    out[4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 8, so - 56, 0));      // 8 + 8 = 16: first byte outside, via soffset
    out[5] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 4, so - 56, 0));      // 4 + 8 = 12: last dword inside
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 0, so - 56, 0));   // bytes 8 .. 23: half inside
    out[6] = v[0]; out[7] = v[1]; out[8] = v[2]; out[9] = v[3];
}
int main()
{
    float *p, *o, h[1024], r[10];
    for (int i = 0; i < 1024; ++i) h[i] = 7.0f;
    hipMalloc(&p, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(p, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, p, o, 64);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("inside %.0f | voffset out %.0f | soffset(sgpr) out %.0f | soffset(const) out %.0f | v8+s8 %.0f | v4+s8 %.0f | b128 at 8: %.0f %.0f %.0f %.0f\n",
           r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9]);
    printf("%s\n", r[2] == 0.0f && r[4] == 0.0f && r[5] == 7.0f ? "the scalar offset IS range-checked on this target" : "the scalar offset is NOT range-checked on this target");
    return 0;
}
