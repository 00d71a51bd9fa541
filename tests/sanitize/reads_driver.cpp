// sanitizer driver for gtx_reads_* and gtx_bam_shrink (test tooling, not part of the product)
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
    // the pre-filter over the same file (whole records: names, qualities, tags)
    gtx_shrink_params par;
    gtx_shrink_params_default(&par);
    par.min_read_len = 20;
    par.min_num_matching = 10;
    char const * chroms[2] = {"chrA", "chr1"};
    int32_t const begins[2] = {100, 9000}, ends[2] = {2000, 40000};
    std::string const out = std::string(argv[a]) + ".shrunk";
    for (int which = 0; which < 2; ++which)
    {
      gtx_shrink_stats st;
      if (gtx_bam_shrink(argv[a], chroms + which, begins + which, ends + which, 1, &par, out.c_str(), &st) == 0)
        std::remove(out.c_str());
    }
  }
  std::printf("ok\n");
  return 0;
}
