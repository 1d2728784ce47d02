// smpl_amd/csrc/rtc_helper.cpp -- smplx_rtc: compiles the kernels against a model-constants header.
//   smplx_rtc <model_const.h> <out.hsaco> [-DNAME ...]
// Started by the library as a child process (specialize.cpp) so that the compiler is the ROCm this package was built
// with, whatever hiprtc/comgr the host program has loaded.  Does not touch the GPU.
#include <cstdio>
#include <fstream>
#include <sstream>

#include "rtc_compile.h"

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: smplx_rtc <model_const.h> <out.hsaco> [-DNAME ...]\n"); return 2; }
    std::ifstream in(argv[1]);
    if (!in) { fprintf(stderr, "smplx_rtc: cannot read %s\n", argv[1]); return 2; }
    std::stringstream ss;
    ss << in.rdbuf();
    std::string defs;
    for (int i = 3; i < argc; ++i) { if (i > 3) defs += " "; defs += argv[i]; }
    if (!defs.empty()) setenv("SMPLX_RTC_DEFINES", defs.c_str(), 1);
    std::vector<char> code;
    std::string why;
    if (!smplx::rtc_compile(ss.str(), code, why)) { fprintf(stderr, "smplx_rtc: %s\n", why.c_str()); return 1; }
    std::ofstream out(argv[2], std::ios::binary);
    out.write(code.data(), (std::streamsize)code.size());
    return out.good() ? 0 : 1;
}
