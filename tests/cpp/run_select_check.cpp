// Brute-force check of youtokentome_amd/csrc/run_select.h (carry arithmetic on lane masks); built and run by tests/test_run_select.py.
#include "run_select.h"
using namespace yttm;
typedef unsigned long long u64;
#include <stdio.h>
#include <stdlib.h>
#include <vector>
int main(){
  srand(1);
  for(int trial=0;trial<200000;trial++){
    int nm=1+rand()%4; int N=64*nm;
    std::vector<int> E(N+1,0), Z(N+2,0), mark(N+1,0);
    int dens=1+rand()%9;
    for(int i=0;i<N;i++){E[i]=(rand()%10)<dens; Z[i]=(rand()%10)<dens; mark[i]=rand()%3==0;}
    // brute force
    std::vector<int> selb(N,0), selzb(N,0), endb(N+1,0);
    for(int i=0;i<N;i++) if(E[i]){ int s=i; while(s>0&&E[s-1]) s--; selb[i]=((i-s)%2==0); }
    for(int i=0;i<N;i++) if(Z[i]){ int s=i; while(s>=2&&Z[s-2]) s-=2; selzb[i]=(((i-s)/2)%2==0); }
    for(int i=0;i<N;i++) if(E[i]&&(i==0||!E[i-1])&&mark[i]){ int e=i; while(e<N&&E[e]) e++; if(e<N) endb[e]=1; }
    RunCarry rc; Chain2Carry cc; bool cons=false; bool prev_top=false;
    for(int m=0;m<nm;m++){
      u64 e=0,z=0,mk=0; for(int b=0;b<64;b++){ if(E[64*m+b]) e|=1ull<<b; if(Z[64*m+b]) z|=1ull<<b; if(mark[64*m+b]) mk|=1ull<<b; }
      u64 s=even_offset_select(e,rc), sz=stride2_select(z,cc), en=marked_run_ends(e,mk,prev_top,cons);
      prev_top=e>>63;
      for(int b=0;b<64;b++){
        if(((s>>b)&1)!=(u64)selb[64*m+b]){printf("sel mismatch trial %d m %d b %d\n",trial,m,b);return 1;}
        if(((sz>>b)&1)!=(u64)selzb[64*m+b]){printf("selz mismatch trial %d m %d b %d\n",trial,m,b);return 1;}
        if(((en>>b)&1)!=(u64)endb[64*m+b]){printf("end mismatch trial %d m %d b %d got %d want %d\n",trial,m,b,(int)((en>>b)&1),endb[64*m+b]);return 1;}
      }
    }
  }
  printf("ok\n");
}
