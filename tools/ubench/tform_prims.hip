// One kernel per T-form primitive (plus 3 - 8 instructions of addressing each) for tools/isa_count.py:
//   python tools/isa_count.py tools/ubench/tform_prims.hip --top 12
#include "ntt2t.cuh"
namespace ola {
__global__ void k_to_u64_weak(const int4* in, u64* out) { T4 x; int4 v = in[threadIdx.x]; x.v[0]=v.x;x.v[1]=v.y;x.v[2]=v.z;x.v[3]=v.w; out[threadIdx.x] = tf_to_u64<false>(x); }
__global__ void k_to_u64_canon(const int4* in, u64* out) { T4 x; int4 v = in[threadIdx.x]; x.v[0]=v.x;x.v[1]=v.y;x.v[2]=v.z;x.v[3]=v.w; out[threadIdx.x] = tf_to_u64<true>(x); }
__global__ void k_tf_mul(const int4* in, const int4* w, int4* out) { T4 x; int4 v = in[threadIdx.x]; x.v[0]=v.x;x.v[1]=v.y;x.v[2]=v.z;x.v[3]=v.w; int4 ww = w[threadIdx.x]; TfTw t; t.w0=ww.x;t.w1=ww.y;t.w2=ww.z; T4 y = tf_mul(x,t); out[threadIdx.x] = make_int4(y.v[0],y.v[1],y.v[2],y.v[3]); }
__global__ void k_loadmul(const u64* in, const u64* w, int4* out) { u64 lo, hi; mul_wide(in[threadIdx.x], w[threadIdx.x], lo, hi); T4 y = tf_from_u128(lo,hi); out[threadIdx.x] = make_int4(y.v[0],y.v[1],y.v[2],y.v[3]); }
__global__ void k_from64(const u64* in, int4* out) { T4 y = tf_from_u64(in[threadIdx.x]); out[threadIdx.x] = make_int4(y.v[0],y.v[1],y.v[2],y.v[3]); }
__global__ void k_dft4(const int4* in, int4* out) { T4 x[16]; for (int j=0;j<16;j++){int4 v = in[threadIdx.x*16+j]; x[j].v[0]=v.x;x[j].v[1]=v.y;x[j].v[2]=v.z;x[j].v[3]=v.w;} tf_dft<4,false>(x); for (int j=0;j<16;j++) out[threadIdx.x*16+j]=make_int4(x[j].v[0],x[j].v[1],x[j].v[2],x[j].v[3]); }
__global__ void k_dft3(const int4* in, int4* out) { T4 x[8]; for (int j=0;j<8;j++){int4 v = in[threadIdx.x*8+j]; x[j].v[0]=v.x;x[j].v[1]=v.y;x[j].v[2]=v.z;x[j].v[3]=v.w;} tf_dft<3,false>(x); for (int j=0;j<8;j++) out[threadIdx.x*8+j]=make_int4(x[j].v[0],x[j].v[1],x[j].v[2],x[j].v[3]); }
__global__ void k_glmul(const u64* in, const u64* w, u64* out) { out[threadIdx.x] = gl_mul(in[threadIdx.x], w[threadIdx.x]); }
__global__ void k_norm(const int4* in, int4* out) { T4 x; int4 v = in[threadIdx.x]; x.v[0]=v.x;x.v[1]=v.y;x.v[2]=v.z;x.v[3]=v.w; T4 y = tf_norm(x); out[threadIdx.x] = make_int4(y.v[0],y.v[1],y.v[2],y.v[3]); }
}
