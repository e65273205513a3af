// Issue-rate microbenchmark for the VALU instructions the scan kernels are built from.
// 6 blocks x 4 waves per CU, 8 independent dependency chains per lane, inline asm so the
// compiler cannot substitute instructions.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHAINS 8
#define BODY(ASM) \
  for (int it = 0; it < iters; it++) { \
    _Pragma("unroll") for (int i = 0; i < CHAINS; i++) { asm volatile(ASM : "+v"(a[i]) : "v"(b[i]), "v"(c) ); } }
template<int OP> __global__ void k(uint32_t* out, int iters){
  uint32_t a[CHAINS], b[CHAINS]; uint32_t c = threadIdx.x * 2654435761u + 12345u;
  for (int i=0;i<CHAINS;i++){ a[i]=c+i*77u; b[i]=c*3+i; }
  if (OP==0)  BODY("v_xor_b32 %0, %0, %1")
  if (OP==1)  BODY("v_pk_min_u16 %0, %0, %1")
  if (OP==2)  BODY("v_pk_sub_u16 %0, %0, %1")
  if (OP==3)  BODY("v_pk_add_u16 %0, %0, %1")
  if (OP==4)  BODY("v_min_u32 %0, %0, %1")
  if (OP==5)  BODY("v_add_u32 %0, %0, %1")
  if (OP==6)  BODY("v_and_or_b32 %0, %0, %1, %2")
  if (OP==7)  BODY("v_bfi_b32 %0, %0, %1, %2")
  if (OP==8)  BODY("v_alignbit_b32 %0, %0, %1, 6")
  if (OP==9)  BODY("v_alignbyte_b32 %0, %0, %1, 1")
  if (OP==10) BODY("v_perm_b32 %0, %0, %1, %2")
  if (OP==11) BODY("v_lshl_or_b32 %0, %0, 3, %1")
  if (OP==12) BODY("v_or3_b32 %0, %0, %1, %2")
  if (OP==13) BODY("v_mul_u32_u24 %0, %0, %1")
  if (OP==14) BODY("v_mad_u32_u24 %0, %0, %1, %2")
  if (OP==15) BODY("v_sad_u16 %0, %0, %1, %2")
  if (OP==16) BODY("v_msad_u8 %0, %0, %1, %2")
  if (OP==17) BODY("v_cmp_eq_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc")
  if (OP==18) BODY("v_cmp_eq_u16_sdwa vcc, %0, %1 src0_sel:WORD_0 src1_sel:WORD_1\n v_addc_co_u32 %0, vcc, 0, %0, vcc")
  if (OP==19) BODY("v_lshrrev_b32 %0, 3, %0")
  if (OP==20) BODY("v_sub_u32 %0, %0, %1")
  if (OP==21) BODY("v_max_u32 %0, %0, %1")
  if (OP==22) BODY("v_and_b32 %0, %0, %1")
  if (OP==23) BODY("v_bfe_u32 %0, %0, 3, 5")
  if (OP==24) BODY("v_lshl_add_u32 %0, %0, 2, %1")
  if (OP==25) BODY("v_min_u16 %0, %0, %1")
  if (OP==26) BODY("v_pk_max_u16 %0, %0, %1")
  if (OP==27) BODY("v_cndmask_b32 %0, %0, %1, vcc")
  uint32_t s=0; for(int i=0;i<CHAINS;i++) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<int OP> float run(uint32_t* d, int iters){ hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); float best=1e9;
  for(int rep=0;rep<3;rep++){ hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(256*6), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); if(ms<best) best=ms; } return best; }
int main(){ uint32_t* d; hipMalloc(&d, 256*6*256*4); const int iters=4000;
 const char* names[]={"v_xor_b32","v_pk_min_u16","v_pk_sub_u16","v_pk_add_u16","v_min_u32","v_add_u32","v_and_or_b32","v_bfi_b32","v_alignbit_b32","v_alignbyte_b32","v_perm_b32","v_lshl_or_b32","v_or3_b32","v_mul_u32_u24","v_mad_u32_u24","v_sad_u16","v_msad_u8","v_cmp_eq_u32+v_addc","v_cmp_eq_u16_sdwa+v_addc","v_lshrrev_b32","v_sub_u32","v_max_u32","v_and_b32","v_bfe_u32","v_lshl_add_u32","v_min_u16","v_pk_max_u16","v_cndmask_b32"};
 float ms[28];
 ms[0]=run<0>(d,iters); ms[1]=run<1>(d,iters); ms[2]=run<2>(d,iters); ms[3]=run<3>(d,iters); ms[4]=run<4>(d,iters); ms[5]=run<5>(d,iters); ms[6]=run<6>(d,iters); ms[7]=run<7>(d,iters);
 ms[8]=run<8>(d,iters); ms[9]=run<9>(d,iters); ms[10]=run<10>(d,iters); ms[11]=run<11>(d,iters); ms[12]=run<12>(d,iters); ms[13]=run<13>(d,iters); ms[14]=run<14>(d,iters); ms[15]=run<15>(d,iters);
 ms[16]=run<16>(d,iters); ms[17]=run<17>(d,iters); ms[18]=run<18>(d,iters); ms[19]=run<19>(d,iters); ms[20]=run<20>(d,iters); ms[21]=run<21>(d,iters); ms[22]=run<22>(d,iters); ms[23]=run<23>(d,iters);
 ms[24]=run<24>(d,iters); ms[25]=run<25>(d,iters); ms[26]=run<26>(d,iters); ms[27]=run<27>(d,iters);
 for(int op=0;op<28;op++){ double n = 256.0*6*4*iters*CHAINS*((op==17||op==18)?2:1); double per_simd = n/(256*4); printf("%-28s %.3f ms  %.2f cycles/instr/SIMD (@2.4GHz nominal)\n", names[op], ms[op], ms[op]*1e-3*2.4e9/per_simd); }
 return 0; }
