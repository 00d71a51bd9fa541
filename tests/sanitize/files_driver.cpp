// sanitizer driver for gtx_graph_from_files (test tooling, not part of the product): argv = directories holding t.fa, t.fa.fai, t.vcf and a file "args" with: region add_all sv
#include "gtx.h"
#include <cstdio>
#include <fstream>
#include <string>
namespace gtx { thread_local std::string g_last_error; }
int main(int argc, char ** argv)
{
  for (int a = 1; a < argc; ++a)
  {
    std::string d = argv[a];
    std::ifstream in(d + "/args");
    std::string region; int aav = 0, sv = 0;
    in >> region >> aav >> sv;
    gtx_graph * g = nullptr;
    int64_t b = 0, e = 0;
    int rc = gtx_graph_from_files((d + "/t.fa").c_str(), (d + "/t.vcf").c_str(), region.c_str(), aav, sv, &g, &b, &e);
    if (rc == 0 && g)
    {
      gtx_graph_view v;
      gtx_graph_get_view(g, &v);
      gtx_graph_destroy(g);
    }
  }
  std::printf("ok\n");
  return 0;
}
