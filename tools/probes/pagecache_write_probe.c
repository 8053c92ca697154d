#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static double now(){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+t.tv_nsec*1e-9;}
#define CH 65536
static size_t total=(size_t)1600<<20; static int nth; static int fd; static char* map; static char* src; static int mode;
static void* run(void* a){long id=(long)a; size_t n=total/CH; for(size_t i=id;i<n;i+=nth){ if(mode==0){ if(pwrite(fd,src+(i%64)*CH,CH,i*CH)!=CH) perror("pw"); } else memcpy(map+i*CH,src+(i%64)*CH,CH);} return 0;}
int main(int argc,char**argv){mode=atoi(argv[1]);nth=atoi(argv[2]);const char* path=argv[3];
 src=malloc(64*CH);memset(src,7,64*CH);
 unlink(path);fd=open(path,O_RDWR|O_CREAT,0644);
 double t0=now();
 if(mode>=1){ if(mode==2){ if(posix_fallocate(fd,0,total)) perror("falloc"); } else if(ftruncate(fd,total)) perror("ftr"); map=mmap(0,total,PROT_READ|PROT_WRITE,MAP_SHARED,fd,0); if(map==MAP_FAILED){perror("mmap");return 1;} }
 double t1=now();
 pthread_t th[64];for(long i=0;i<nth;i++)pthread_create(&th[i],0,run,(void*)i);for(int i=0;i<nth;i++)pthread_join(th[i],0);
 double t2=now();
 printf("mode %d threads %d: setup %.1f ms, write %.1f ms = %.2f GB/s\n",mode,nth,(t1-t0)*1e3,(t2-t1)*1e3,total/(t2-t1)/1e9);
 close(fd);unlink(path);return 0;}
