#include <charconv>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <random>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include "../../visgeom_amd/csrc/vg_host_parallel.hpp"
static void app(std::string &o, const double *v, int n){ char b[40]; for(int i=0;i<n;i++){ auto r=std::to_chars(b,b+40,v[i],std::chars_format::general,6); if(i) o.push_back(' '); o.append(b,r.ptr-b);} }
int main(){
  const size_t n=10000; const int N=96;
  std::vector<double> pr(2*N*n), det(2*N*n), xi(6*n);
  std::mt19937_64 g(1); std::uniform_real_distribution<double> U(0,1280);
  for(auto&x:pr)x=U(g); for(size_t i=0;i<det.size();i++)det[i]=pr[i]+U(g)*1e-3; for(auto&x:xi)x=U(g)*1e-3;
  for(int rep=0;rep<3;rep++){
  auto t0=std::chrono::steady_clock::now();
  std::vector<std::string> text(vgpar::host_threads());
  int parts=vgpar::parallel_ranges(n,64,[&](size_t b,size_t e,int part){ std::string&out=text[part]; out.reserve((e-b)*N*100);
    for(size_t k=b;k<e;k++){ std::string pose="   "; app(pose,&xi[6*k],3); pose+=" "; app(pose,&xi[6*k+3],3); pose+="\n";
      for(int i=0;i<N;i++){ double err[2]={det[2*N*k+2*i]-pr[2*N*k+2*i],det[2*N*k+2*i+1]-pr[2*N*k+2*i+1]}; app(out,err,2); out+="   "; app(out,&pr[2*N*k+2*i],2); out+=pose; } } });
  auto t1=std::chrono::steady_clock::now();
  size_t tot=0; for(auto&s:text)tot+=s.size();
  FILE*f=fopen("/tmp/vg_probe_out1.txt","wb"); for(int k=0;k<parts;k++)fwrite(text[k].data(),1,text[k].size(),f); fclose(f);
  auto t2=std::chrono::steady_clock::now();
  int fd=open("/tmp/vg_probe_out2.txt",O_RDWR|O_CREAT|O_TRUNC,0644); ftruncate(fd,tot); char*m=(char*)mmap(nullptr,tot,PROT_WRITE,MAP_SHARED,fd,0);
  std::vector<size_t> off(parts+1,0); for(int k=0;k<parts;k++)off[k+1]=off[k]+text[k].size();
  vgpar::parallel_ranges(parts,1,[&](size_t b,size_t e,int){ for(size_t k=b;k<e;k++) memcpy(m+off[k],text[k].data(),text[k].size()); });
  munmap(m,tot); close(fd);
  auto t3=std::chrono::steady_clock::now();
  fd=open("/tmp/vg_probe_out3.txt",O_RDWR|O_CREAT|O_TRUNC,0644);
  vgpar::parallel_ranges(parts,1,[&](size_t b,size_t e,int){ for(size_t k=b;k<e;k++){ size_t done=0; while(done<text[k].size()){ ssize_t w=pwrite(fd,text[k].data()+done,text[k].size()-done,off[k]+done); if(w<=0)break; done+=w;} } });
  close(fd);
  auto t4=std::chrono::steady_clock::now();
  auto d=[](auto a,auto b){return std::chrono::duration<double>(b-a).count()*1e3;};
  printf("threads %d bytes %zu format %.1f ms fwrite %.1f ms mmap %.1f ms pwrite %.1f ms\n",parts,tot,d(t0,t1),d(t1,t2),d(t2,t3),d(t3,t4));
  }
}
