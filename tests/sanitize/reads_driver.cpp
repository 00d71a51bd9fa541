// sanitizer driver for gtx_reads_* (test tooling, not part of the product)
#include "gtx.h"
#include <cstdio>
#include <cstdint>
#include <string>
#include <vector>
namespace gtx { thread_local std::string g_last_error; }
int main(int argc, char ** argv)
{
  for (int a = 1; a < argc; ++a)
  {
    for (int with_region = 0; with_region < 2; ++with_region)
    {
      char const * paths[1] = {argv[a]};
      gtx_reads * r = nullptr;
      int rc = gtx_reads_open(paths, 1, with_region ? "chrA:100-2000" : nullptr, &r);
      if (rc != 0)
        continue;
      uint32_t ns = 0, nrg = 0;
      gtx_reads_info(r, &ns, &nrg);
      for (uint32_t i = 0; i < ns; ++i)
        (void)gtx_reads_sample_name(r, i);
      std::vector<gtx_stream_record> recs(64);
      std::vector<uint8_t> seq(64 * 80);
      uint32_t n = 0;
      long total = 0;
      while (gtx_reads_next(r, recs.data(), seq.data(), 80, 64, &n) == 0 && n != 0)
        total += n;
      gtx_reads_close(r);
    }
  }
  std::printf("ok\n");
  return 0;
}
