#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pkmin(uint32_t a, uint32_t b){ u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(u16x2,a), __builtin_bit_cast(u16x2,b)); return __builtin_bit_cast(uint32_t,r); }
template<int OP> __global__ void k(uint32_t* out, int iters){
  uint32_t a[8]; uint32_t x = threadIdx.x*2654435761u + blockIdx.x;
  for(int i=0;i<8;i++) a[i]=x+i*77u;
  for(int it=0; it<iters; it++){
#pragma unroll
    for(int i=0;i<8;i++){
      if(OP==0) a[i] = pkmin(a[i], a[(i+1)&7] ^ (uint32_t)it);
      if(OP==1) a[i] = a[i] & (a[(i+1)&7] ^ (uint32_t)it);
      if(OP==2) a[i] = __builtin_amdgcn_alignbit(a[i], a[(i+1)&7], 6) ^ (uint32_t)it;
      if(OP==3) a[i] = min(a[i], a[(i+1)&7] ^ (uint32_t)it);
    }
  }
  uint32_t s=0; for(int i=0;i<8;i++) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
int main(){ uint32_t* d; hipMalloc(&d, 256*24*256*4); hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
 const char* names[4]={"pk_min_u16+xor","and+xor","alignbit+xor","min_u32+xor"};
 for(int op=0;op<4;op++){ for(int rep=0;rep<2;rep++){ hipEventRecord(e0);
   int iters=2000; dim3 g(256*6), b(256);
   if(op==0) hipLaunchKernelGGL(k<0>,g,b,0,0,d,iters); if(op==1) hipLaunchKernelGGL(k<1>,g,b,0,0,d,iters); if(op==2) hipLaunchKernelGGL(k<2>,g,b,0,0,d,iters); if(op==3) hipLaunchKernelGGL(k<3>,g,b,0,0,d,iters);
   hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1);
   double instr = 256.0*6*4 /*waves*/ * iters * 8 * 2; // 2 ops per element
   if(rep) printf("%s: %.3f ms, %.2f G wave-instr/s, %.2f cycles/instr/SIMD @2.4GHz\n", names[op], ms, instr/ms/1e6, 2.4e9/(instr/(ms*1e-3)/(256*4))); } }
 return 0; }
