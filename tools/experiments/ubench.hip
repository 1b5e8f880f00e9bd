#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template<int OP> __global__ void k(double* out, double seed){
  double a0=seed+threadIdx.x*1e-3, a1=a0+1, a2=a0+2, a3=a0+3, a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
  double m=1.0000001, c=1e-9;
  for(int i=0;i<ITER;i++){
    if(OP==0){ a0=fma(a0,m,c);a1=fma(a1,m,c);a2=fma(a2,m,c);a3=fma(a3,m,c);a4=fma(a4,m,c);a5=fma(a5,m,c);a6=fma(a6,m,c);a7=fma(a7,m,c);}   
    if(OP==1){ a0=__builtin_amdgcn_rcp(a0);a1=__builtin_amdgcn_rcp(a1);a2=__builtin_amdgcn_rcp(a2);a3=__builtin_amdgcn_rcp(a3);a4=__builtin_amdgcn_rcp(a4);a5=__builtin_amdgcn_rcp(a5);a6=__builtin_amdgcn_rcp(a6);a7=__builtin_amdgcn_rcp(a7);}
    if(OP==2){ a0=a0*m;a1=a1*m;a2=a2*m;a3=a3*m;a4=a4*m;a5=a5*m;a6=a6*m;a7=a7*m;}
    if(OP==3){ a0=a0+c;a1=a1+c;a2=a2+c;a3=a3+c;a4=a4+c;a5=a5+c;a6=a6+c;a7=a7+c;}
    if(OP==4){ a0=fma(a0,m,c);a0=fma(a0,m,c);a0=fma(a0,m,c);a0=fma(a0,m,c);a0=fma(a0,m,c);a0=fma(a0,m,c);a0=fma(a0,m,c);a0=fma(a0,m,c);} // dependent chain
    if(OP==5){ a0=sqrt(a0);a1=sqrt(a1);a2=sqrt(a2);a3=sqrt(a3);a4=sqrt(a4);a5=sqrt(a5);a6=sqrt(a6);a7=sqrt(a7);}
    if(OP==6){ float f0=(float)a0,f1=(float)a1,f2=(float)a2,f3=(float)a3; f0=fmaf(f0,1.0001f,1e-5f);f1=fmaf(f1,1.0001f,1e-5f);f2=fmaf(f2,1.0001f,1e-5f);f3=fmaf(f3,1.0001f,1e-5f);f0=fmaf(f0,1.0001f,1e-5f);f1=fmaf(f1,1.0001f,1e-5f);f2=fmaf(f2,1.0001f,1e-5f);f3=fmaf(f3,1.0001f,1e-5f); a0=f0;a1=f1;a2=f2;a3=f3;}
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=a0+a1+a2+a3+a4+a5+a6+a7;
}
template<int OP> void run(const char* name, int waves_per_simd){
  double* d; hipMalloc(&d, 1<<24);
  int blocks=256*4*waves_per_simd; // 64-thread blocks
  hipEvent_t a,b; hipEventCreate(&a);hipEventCreate(&b);
  k<OP><<<blocks,64>>>(d,1.5); hipDeviceSynchronize();
  hipEventRecord(a); k<OP><<<blocks,64>>>(d,1.5); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b);
  double instr_per_simd = (double)waves_per_simd*ITER*8;
  printf("%-10s waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instr per SIMD (at 2.4GHz)\n", name, waves_per_simd, ms, ms*1e-3*2.4e9/instr_per_simd);
  hipFree(d);
}
int main(){ for(int w: {1,2,4}){ run<0>("fma",w); run<1>("rcp",w); run<2>("mul",w); run<3>("add",w); run<4>("fma-dep",w); run<5>("sqrt",w);} return 0; }
