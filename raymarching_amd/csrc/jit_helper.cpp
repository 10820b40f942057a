// jit_helper.cpp -- rm_jit: compiles one kernel source with hiprtc in a process of its own.
//
//   rm_jit <digest of the device headers> <source file> <code object file> [compiler option ...]
//
// The library starts it for every scene whose kernels it has to compile (jit.cpp).  A process that holds other ROCm
// libraries -- PyTorch ships its own hiprtc and code-object manager under the same SONAMEs, an older release than the
// system's -- would otherwise compile the kernels with whichever copy the dynamic loader met first; this program is linked
// against the ROCm installation it was built with, so the kernels are always that compiler's (measured: 3 % faster
// on the shipped frame, 6 % on a 1024-primitive scene, than the release bundled with PyTorch 2.10 makes them).
// Exit status: 0 = code object written; 2 = the compiler refused the source, its log is in "<code object file>.log";
// 11 = this program was built with other device headers than the library that started it (the digest differs: one of
// the two is stale); anything else = the program could not do its work.  The library then compiles in its own process.
#include <hip/hiprtc.h>

#include <cstdio>
#include <string>
#include <vector>

namespace rmdev {
struct EmbeddedHeader { const char *name; const char *text; };
extern const EmbeddedHeader kEmbeddedHeaders[];
extern const unsigned kEmbeddedHeaderCount;
}  // namespace rmdev

namespace {

bool read_file(const char *path, std::string &out) {
    std::FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    const bool ok = !std::ferror(f);
    std::fclose(f);
    return ok;
}

bool write_file(const std::string &path, const char *data, size_t bytes) {
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(data, 1, bytes, f) == bytes;
    return std::fclose(f) == 0 && ok;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: rm_jit <digest of the device headers> <source file> <code object file> [compiler option ...]\n");
        return 3;
    }
    {
        // the headers compiled into this program must be the ones the library generated the source for (FNV-1a, as jit.cpp)
        unsigned long long h = 1469598103934665603ull;
        for (unsigned i = 0; i < rmdev::kEmbeddedHeaderCount; i++)
            for (const char *text : { rmdev::kEmbeddedHeaders[i].name, rmdev::kEmbeddedHeaders[i].text })
                for (const char *c = text; *c; ++c) h = (h ^ (unsigned char)*c) * 1099511628211ull;
        char mine[32];
        std::snprintf(mine, sizeof mine, "%016llx", h);
        if (std::string(mine) != argv[1]) return 11;
    }
    argv++;
    argc--;
    std::string source;
    if (!read_file(argv[1], source)) return 4;
    std::vector<const char *> texts, names;
    for (unsigned i = 0; i < rmdev::kEmbeddedHeaderCount; i++) {
        texts.push_back(rmdev::kEmbeddedHeaders[i].text);
        names.push_back(rmdev::kEmbeddedHeaders[i].name);
    }
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, source.c_str(), "scene_kernels.hip", (int)texts.size(), texts.data(), names.data()) != HIPRTC_SUCCESS) return 5;
    const hiprtcResult r = hiprtcCompileProgram(prog, argc - 3, const_cast<const char **>(argv + 3));
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        (void)hiprtcGetProgramLogSize(prog, &n);
        std::string text(n, '\0');
        if (n) (void)hiprtcGetProgramLog(prog, &text[0]);
        const std::string log = std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(r) + "\n" + text;
        return write_file(std::string(argv[2]) + ".log", log.data(), log.size()) ? 2 : 6;
    }
    size_t size = 0;
    if (hiprtcGetCodeSize(prog, &size) != HIPRTC_SUCCESS || size == 0) return 7;
    std::vector<char> code(size);
    if (hiprtcGetCode(prog, code.data()) != HIPRTC_SUCCESS) return 8;
    (void)hiprtcDestroyProgram(&prog);
    // complete or absent: a caller that cannot learn this program's exit status (SIGCHLD ignored) goes by the file
    const std::string part = std::string(argv[2]) + ".part";
    if (!write_file(part, code.data(), code.size())) return 9;
    return std::rename(part.c_str(), argv[2]) == 0 ? 0 : 10;
}
