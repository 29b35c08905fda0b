// exhaustive check: 5-op constant division == IEEE division for all floats a
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static inline float divc(float a, float b, float y){
  float q=a*y; float r=fmaf(-b,q,a); q=fmaf(r,y,q); r=fmaf(-b,q,a); return fmaf(r,y,q);
}
static inline float divc1(float a, float b, float y){
  float q=a*y; float r=fmaf(-b,q,a); return fmaf(r,y,q);
}
int main(int argc,char**argv){
  float b=atof(argv[1]); float y=(float)(1.0/(double)b);
  // correctly rounded reciprocal: check neighbours in long double
  long double t=1.0L/(long double)b; float best=y; 
  for(int d=-1;d<=1;d++){ uint32_t u; memcpy(&u,&y,4); u+=d; float c; memcpy(&c,&u,4); if(fabsl((long double)c-t)<fabsl((long double)best-t)) best=c; }
  y=best;
  uint32_t lo=strtoul(argv[2],0,16), hi=strtoul(argv[3],0,16);
  unsigned long bad5=0,bad3=0,n=0; float ex=0;
  #pragma omp parallel for reduction(+:bad5,bad3,n)
  for(uint64_t u=lo;u<=hi;u++){ uint32_t v=(uint32_t)u; float a; memcpy(&a,&v,4); if(a!=a||isinf(a)) continue; float ref=a/b; float g=divc(a,b,y); float g1=divc1(a,b,y); n++;
    if(memcmp(&ref,&g,4)!=0 && !(ref==0&&g==0)) { bad5++; if(bad5<3) printf("bad5 a=%a ref=%a got=%a\n",a,ref,g);} if(memcmp(&ref,&g1,4)!=0 && !(ref==0&&g1==0)) bad3++; }
  printf("b=%g y=%a n=%lu bad5=%lu bad3=%lu\n",b,y,n,bad5,bad3);
}
