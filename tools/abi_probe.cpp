#include <cstdio>
#include <dlfcn.h>
int main(int argc,char**argv){ void*h=dlopen(argv[1],RTLD_NOW); if(!h){printf("dlopen fail %s\n",dlerror());return 1;}
 auto f=(int(*)())dlsym(h,"vstab_device_count"); auto e=(const char*(*)())dlsym(h,"vstab_last_error"); int n=f(); printf("devices=%d err=%s\n",n,e()); return 0;}
