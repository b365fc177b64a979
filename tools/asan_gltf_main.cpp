#include <cstdio>
#include "pt_api.h"
int main(int argc, char** argv)
{
  int ok = 0, bad = 0;
  for(int i = 1; i < argc; ++i)
  {
    pt_GltfScene* s = nullptr;
    char          err[256];
    if(pt_gltf_load(argv[i], &s, err, sizeof(err)) == PT_OK) { ++ok; pt_gltf_free(s); }
    else ++bad;
  }
  std::printf("ok %d rejected %d\n", ok, bad);
  return 0;
}
