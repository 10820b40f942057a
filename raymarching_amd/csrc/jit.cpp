// jit.cpp -- compiles the kernel source generated for a scene with hiprtc and launches it.
//
// Same compiler flags as the ahead-of-time kernels (Makefile): -ffp-contract=off (one rounding
// per fp32 operation), no SLP packing, correctly rounded sqrt and divide.
#include "jit.h"

#include <hip/hiprtc.h>

#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <cstdlib>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace rmdev {

namespace {

using CodeObject = std::shared_ptr<const std::vector<char>>;
std::mutex g_cache_lock;
std::unordered_map<std::string, CodeObject> g_cache;   // source text and $RM_JIT_EXTRA -> gfx950 code object
// compiles started ahead of their use (precompile_scene_source), by the same key; declared behind the cache they fill, so
// that at the end of the process they are waited for while it still exists
struct Compiled { CodeObject code; std::string log; };
std::unordered_map<std::string, std::shared_future<Compiled>> g_in_flight;
std::string g_cache_dir;                                // second level on disk ("" = none)
bool g_cache_dir_chosen = false;                        // set_kernel_cache_dir was called, or the default was looked up

// A directory of the calling user's that can be created and written: the cache's default place.
bool usable_dir(const std::string &dir) {
    if (dir.empty()) return false;
    std::string at;
    for (size_t i = 0; i <= dir.size(); i++) {            // mkdir -p, directories private to the user
        if (i == dir.size() || (dir[i] == '/' && i > 0)) {
            at = dir.substr(0, i);
            if (mkdir(at.c_str(), 0700) != 0 && errno != EEXIST) return false;
        }
    }
    struct stat st;
    if (stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
    if (st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return false;   // never load code another user could have put there
    return access(dir.c_str(), W_OK | X_OK) == 0;
}

// $RM_KERNEL_CACHE_DIR ("" or "off": none), else $XDG_CACHE_HOME/raymarch_hip, $HOME/.cache/raymarch_hip, /tmp/raymarch_hip-<uid>.
std::string default_cache_dir() {
    if (const char *env = std::getenv("RM_KERNEL_CACHE_DIR")) {
        const std::string v = env;
        if (v.empty() || v == "off") return "";
        return usable_dir(v) ? v : "";
    }
    std::vector<std::string> candidates;
    if (const char *xdg = std::getenv("XDG_CACHE_HOME")) if (*xdg) candidates.push_back(std::string(xdg) + "/raymarch_hip");
    if (const char *home = std::getenv("HOME")) if (*home) candidates.push_back(std::string(home) + "/.cache/raymarch_hip");
    candidates.push_back("/tmp/raymarch_hip-" + std::to_string((long long)geteuid()));
    for (const std::string &c : candidates)
        if (usable_dir(c)) return c;
    return "";
}

// $RM_JIT_EXTRA: further compiler options, separated by blanks (for sweeps of code generation settings; part of the
// cache key, never of the arithmetic: -ffp-contract=off and the rounding options above stay)
std::vector<std::string> extra_options() {
    std::vector<std::string> out;
    if (const char *env = std::getenv("RM_JIT_EXTRA")) {
        std::string word;
        for (const char *c = env;; c++) {
            if (*c == ' ' || *c == '\0') {
                if (!word.empty()) out.push_back(word);
                word.clear();
                if (*c == '\0') break;
            } else {
                word += *c;
            }
        }
    }
    return out;
}

// ---- the compiler in a process of its own ---------------------------------------------------------
// rm_jit (jit_helper.cpp), beside this library: $RM_JIT_HELPER names another program, "" or "off" keeps the compiler in
// this process.  "" if there is none.
extern "C" char **environ;

std::string find_helper() {
    if (const char *env = std::getenv("RM_JIT_HELPER")) {
        const std::string v = env;
        if (v.empty() || v == "off") return "";
        return access(v.c_str(), X_OK) == 0 ? v : "";
    }
    Dl_info info{};
    if (!dladdr(reinterpret_cast<const void *>(&find_helper), &info) || !info.dli_fname) return "";
    std::string path = info.dli_fname;
    const size_t slash = path.rfind('/');
    path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/rm_jit";
    return access(path.c_str(), X_OK) == 0 ? path : "";
}

// (Never destroyed, like the other texts below that a background compile may still read while the process ends.)
const std::string &helper_path() {
    static const std::string &path = *new std::string(find_helper());
    return path;
}

std::string &g_compiler_used = *new std::string("none yet");   // how this process last obtained a code object (kernel_compiler_used)

bool read_whole_file(const std::string &path, std::vector<char> &out) {
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    const bool ok = !std::ferror(f);
    std::fclose(f);
    return ok;
}

// Compiles `source` with the helper program.  1: `code` holds the code object; 0: the compiler refused the source (`log`);
// -1: the helper could not be used (the caller compiles in this process).
int compile_with_helper(const std::string &source, const std::vector<const char *> &options, std::vector<char> &code, std::string &log) {
    const std::string &helper = helper_path();
    if (helper.empty()) return -1;
    const char *tmp = std::getenv("TMPDIR");
    std::string src_path = std::string(tmp && *tmp ? tmp : "/tmp") + "/rm_jit_XXXXXX";
    const int fd = mkstemp(&src_path[0]);   // 0600, a name nobody can guess
    if (fd < 0) return -1;
    const std::string out_path = src_path + ".bin", log_path = out_path + ".log";
    auto cleanup = [&] {
        (void)unlink(src_path.c_str());
        (void)unlink(out_path.c_str());
        (void)unlink((out_path + ".part").c_str());
        (void)unlink(log_path.c_str());
    };
    size_t written = 0;
    while (written < source.size()) {
        const ssize_t n = write(fd, source.data() + written, source.size() - written);
        if (n <= 0) break;
        written += (size_t)n;
    }
    (void)close(fd);
    if (written != source.size()) { cleanup(); return -1; }
    // the program checks that it was built with the device headers this library holds
    static const std::string &digest = *new std::string([] {
        unsigned long long h = 1469598103934665603ull;
        for (unsigned i = 0; i < kEmbeddedHeaderCount; i++)
            for (const char *text : { kEmbeddedHeaders[i].name, kEmbeddedHeaders[i].text })
                for (const char *c = text; *c; ++c) h = (h ^ (unsigned char)*c) * 1099511628211ull;
        char hex[32];
        std::snprintf(hex, sizeof hex, "%016llx", h);
        return std::string(hex);
    }());
    std::vector<char *> argv = { const_cast<char *>(helper.c_str()), const_cast<char *>(digest.c_str()), const_cast<char *>(src_path.c_str()),
                                 const_cast<char *>(out_path.c_str()) };
    for (const char *o : options) argv.push_back(const_cast<char *>(o));
    argv.push_back(nullptr);
    pid_t pid = 0;
    if (posix_spawn(&pid, helper.c_str(), nullptr, nullptr, argv.data(), environ) != 0) { cleanup(); return -1; }
    int status = 0;
    bool reaped = true;
    while (waitpid(pid, &status, 0) < 0) {
        if (errno == EINTR) continue;
        // ECHILD: the calling program ignores SIGCHLD and the child is gone without a status; its code object file is
        // complete if it exists (the program renames it into place)
        reaped = false;
        break;
    }
    int result = -1;
    if (!reaped) {
        code.clear();
        if (errno == ECHILD && read_whole_file(out_path, code) && !code.empty()) result = 1;
    } else if (WIFEXITED(status) && WEXITSTATUS(status) == 0) {
        code.clear();
        if (read_whole_file(out_path, code) && !code.empty()) result = 1;
    } else if (WIFEXITED(status) && WEXITSTATUS(status) == 2) {
        std::vector<char> text;
        if (read_whole_file(log_path, text)) {
            log.assign(text.begin(), text.end());
            result = 0;
        }
    }
    cleanup();
    return result;
}

// FNV-1a over the source and the embedded headers: names a code object file.  The file also holds
// the full key text, compared on load, so a hash collision costs a recompile, never a wrong kernel.
std::string cache_key_text(const std::string &source) {
    std::string key = source;
    for (const std::string &o : extra_options()) key += "\n//@option " + o;
    if (!helper_path().empty()) key += "\n//@compiler rm_jit";   // (objects this process's own hiprtc made are kept apart)
    for (unsigned i = 0; i < kEmbeddedHeaderCount; i++) {
        key += "\n//@";
        key += kEmbeddedHeaders[i].name;
        key += "\n";
        key += kEmbeddedHeaders[i].text;
    }
    return key;
}

std::string cache_dir() {
    std::lock_guard<std::mutex> hold(g_cache_lock);
    if (!g_cache_dir_chosen) {
        g_cache_dir = default_cache_dir();
        g_cache_dir_chosen = true;
    }
    return g_cache_dir;
}

std::string cache_file(const std::string &dir, const std::string &key) {
    unsigned long long h = 1469598103934665603ull;
    for (unsigned char c : key) h = (h ^ c) * 1099511628211ull;
    char name[64];
    std::snprintf(name, sizeof name, "/rm_%016llx.gfx950.bin", h);
    return dir + name;
}

// A file that was just read counts as used now: eviction goes by modification time.
void mark_used(const std::string &path) { (void)utimes(path.c_str(), nullptr); }

// Keeps the directory's rm_*.gfx950.bin files below $RM_KERNEL_CACHE_MAX_MB (default 256): after a store, the files
// unused for longest go until three quarters of the limit are left.  Only files this library names are touched.
void evict_old_files(const std::string &dir) {
    unsigned long long limit = 256ull << 20;
    if (const char *env = std::getenv("RM_KERNEL_CACHE_MAX_MB")) limit = std::strtoull(env, nullptr, 10) << 20;
    DIR *d = opendir(dir.c_str());
    if (!d) return;
    struct Entry { std::string path; unsigned long long bytes; long long used; };
    std::vector<Entry> files;
    unsigned long long total = 0;
    while (const dirent *e = readdir(d)) {
        const size_t n = std::strlen(e->d_name);
        if (n < 15 || std::strncmp(e->d_name, "rm_", 3) != 0 || std::strcmp(e->d_name + n - 11, ".gfx950.bin") != 0) continue;
        const std::string path = dir + "/" + e->d_name;
        struct stat st;
        if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) continue;
        files.push_back({ path, (unsigned long long)st.st_size, (long long)st.st_mtime });
        total += (unsigned long long)st.st_size;
    }
    closedir(d);
    if (total <= limit) return;
    std::sort(files.begin(), files.end(), [](const Entry &a, const Entry &b) { return a.used < b.used; });
    for (const Entry &f : files) {
        if (total <= limit / 4u * 3u) break;
        if (std::remove(f.path.c_str()) == 0) total -= f.bytes;
    }
}

CodeObject load_from_disk(const std::string &source) {
    const std::string dir = cache_dir();
    if (dir.empty()) return nullptr;
    const std::string key = cache_key_text(source);
    const std::string path = cache_file(dir, key);
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return nullptr;
    unsigned long long key_size = 0, code_size = 0;
    CodeObject result;
    if (std::fread(&key_size, sizeof key_size, 1, f) == 1 && std::fread(&code_size, sizeof code_size, 1, f) == 1 &&
        key_size == key.size() && code_size > 0 && code_size < (1ull << 30)) {
        std::string stored(key_size, '\0');
        auto code = std::make_shared<std::vector<char>>(code_size);
        if (std::fread(&stored[0], 1, key_size, f) == key_size && stored == key &&
            std::fread(code->data(), 1, code_size, f) == code_size)
            result = code;
    }
    std::fclose(f);
    if (result) mark_used(path);
    return result;
}

void store_to_disk(const std::string &source, const std::vector<char> &code) {
    const std::string dir = cache_dir();
    if (dir.empty()) return;
    const std::string key = cache_key_text(source), path = cache_file(dir, key), tmp = path + ".tmp" + std::to_string((long long)getpid());
    std::FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return;   // an unwritable cache directory only costs the next process a compile
    const unsigned long long key_size = key.size(), code_size = code.size();
    const bool ok = std::fwrite(&key_size, sizeof key_size, 1, f) == 1 && std::fwrite(&code_size, sizeof code_size, 1, f) == 1 &&
                    std::fwrite(key.data(), 1, key.size(), f) == key.size() && std::fwrite(code.data(), 1, code.size(), f) == code.size();
    std::fclose(f);
    if (ok) std::rename(tmp.c_str(), path.c_str());   // atomic: readers see a whole file or none
    else std::remove(tmp.c_str());
    if (ok) evict_old_files(dir);
}

CodeObject build_code_object(const std::string &source, std::string &log) {
    std::vector<const char *> options = { "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math",
                                          "-fno-slp-vectorize", "-fhip-fp32-correctly-rounded-divide-sqrt" };
    // a part of a split source may ask for a lighter optimisation level (split_parts)
    std::string level;
    {
        const size_t at = source.find("//@optimise ");
        if (at != std::string::npos && (at == 0 || source[at - 1] == '\n')) {
            const size_t end = source.find('\n', at);
            level = source.substr(at + 12, end == std::string::npos ? std::string::npos : end - at - 12);
        }
    }
    if (!level.empty())
        for (const char *&o : options)
            if (std::strcmp(o, "-O3") == 0) o = level.c_str();
    const std::vector<std::string> extra = extra_options();
    for (const std::string &o : extra) options.push_back(o.c_str());
    {
        auto code = std::make_shared<std::vector<char>>();
        const int by_helper = compile_with_helper(source, options, *code, log);
        if (by_helper == 0) return nullptr;
        if (by_helper == 1) {
            std::lock_guard<std::mutex> hold(g_cache_lock);
            g_compiler_used = "rm_jit, a process of its own: " + helper_path();
            return code;
        }
    }
    {
        std::lock_guard<std::mutex> hold(g_cache_lock);
        g_compiler_used = "hiprtc in this process";
    }
    std::vector<const char *> texts, names;
    for (unsigned i = 0; i < kEmbeddedHeaderCount; i++) {
        texts.push_back(kEmbeddedHeaders[i].text);
        names.push_back(kEmbeddedHeaders[i].name);
    }
    hiprtcProgram prog = nullptr;
    hiprtcResult r = hiprtcCreateProgram(&prog, source.c_str(), "scene_kernels.hip", (int)texts.size(), texts.data(), names.data());
    if (r != HIPRTC_SUCCESS) {
        log = std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(r);
        return nullptr;
    }
    r = hiprtcCompileProgram(prog, (int)options.size(), options.data());
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        (void)hiprtcGetProgramLogSize(prog, &n);
        std::string text(n, '\0');
        if (n) (void)hiprtcGetProgramLog(prog, &text[0]);
        log = std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(r) + "\n" + text;
        (void)hiprtcDestroyProgram(&prog);
        return nullptr;
    }
    size_t size = 0;
    (void)hiprtcGetCodeSize(prog, &size);
    auto code = std::make_shared<std::vector<char>>(size);
    r = size ? hiprtcGetCode(prog, code->data()) : HIPRTC_ERROR_INTERNAL_ERROR;
    (void)hiprtcDestroyProgram(&prog);
    if (r != HIPRTC_SUCCESS) {
        log = std::string("hiprtcGetCode: ") + hiprtcGetErrorString(r);
        return nullptr;
    }
    return code;
}

}  // namespace

std::string kernel_compiler_used() {
    std::lock_guard<std::mutex> hold(g_cache_lock);
    return g_compiler_used;
}

void SceneModule::unload() {
    if (module) (void)hipModuleUnload(module);
    for (hipModule_t m : more_modules)
        if (m) (void)hipModuleUnload(m);
    *this = SceneModule();
}

// ---- small keyed blobs beside the code objects (tile orders) ------------------------------------

bool load_cached_blob(const std::string &kind, const std::string &key, std::vector<uint32_t> &words) {
    const std::string dir = cache_dir();
    if (dir.empty()) return false;
    const std::string path = cache_file(dir, kind + key);
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    unsigned long long key_size = 0, n_words = 0;
    bool ok = std::fread(&key_size, sizeof key_size, 1, f) == 1 && std::fread(&n_words, sizeof n_words, 1, f) == 1 &&
              key_size == kind.size() + key.size() && n_words < (1ull << 28);
    if (ok) {
        std::string stored(key_size, '\0');
        words.resize(n_words);
        ok = std::fread(&stored[0], 1, key_size, f) == key_size && stored == kind + key &&
             (n_words == 0 || std::fread(words.data(), sizeof(uint32_t), n_words, f) == n_words);
    }
    std::fclose(f);
    if (ok) mark_used(path);
    return ok;
}

void store_cached_blob(const std::string &kind, const std::string &key, const std::vector<uint32_t> &words) {
    const std::string dir = cache_dir();
    if (dir.empty()) return;
    const std::string full = kind + key, path = cache_file(dir, full), tmp = path + ".tmp" + std::to_string((long long)getpid());
    std::FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return;
    const unsigned long long key_size = full.size(), n_words = words.size();
    const bool ok = std::fwrite(&key_size, sizeof key_size, 1, f) == 1 && std::fwrite(&n_words, sizeof n_words, 1, f) == 1 &&
                    std::fwrite(full.data(), 1, full.size(), f) == full.size() &&
                    (words.empty() || std::fwrite(words.data(), sizeof(uint32_t), words.size(), f) == words.size());
    std::fclose(f);
    if (ok) std::rename(tmp.c_str(), path.c_str());
    else std::remove(tmp.c_str());
    if (ok) evict_old_files(dir);
}

void set_kernel_cache_dir(const char *dir) {
    std::lock_guard<std::mutex> hold(g_cache_lock);
    g_cache_dir = dir ? dir : "";
    g_cache_dir_chosen = true;
}

bool check_scene_source(const std::string &source, std::string &log) {
    if (load_from_disk(source)) return true;
    const CodeObject code = build_code_object(source, log);
    if (code) store_to_disk(source, *code);
    return code != nullptr;
}

namespace {

std::string memory_key_of(const std::string &source) {
    std::string key = source;   // (the embedded headers cannot change within a process; $RM_JIT_EXTRA can)
    for (const std::string &o : extra_options()) key += "\n//@option " + o;
    return key;
}

// The code object of `source` from the cache directory or the compiler, into this process's cache.
CodeObject compile_now(const std::string &source, const std::string &memory_key, std::string &log) {
    CodeObject code = load_from_disk(source);
    if (code) {
        std::lock_guard<std::mutex> hold(g_cache_lock);
        g_compiler_used = "none: code object from the cache directory";
    }
    if (!code) {
        code = build_code_object(source, log);
        if (!code) return nullptr;
        store_to_disk(source, *code);
    }
    std::lock_guard<std::mutex> hold(g_cache_lock);
    if (g_cache.size() >= 256) g_cache.clear();   // a caller that streams ever-new scenes should not grow without bound
    g_cache.emplace(memory_key, code);
    return code;
}

// The code object of `source`: from this process's cache, from a compile started ahead of time, from the cache directory, or
// from the compiler.  `seconds`: what that took (0 from the process's cache).
CodeObject obtain_code(const std::string &source, std::string &log, double &seconds) {
    const std::string memory_key = memory_key_of(source);
    std::shared_future<Compiled> started;
    {
        std::lock_guard<std::mutex> hold(g_cache_lock);
        auto it = g_cache.find(memory_key);
        if (it != g_cache.end()) {
            seconds = 0.0;
            return it->second;
        }
        auto f = g_in_flight.find(memory_key);
        if (f != g_in_flight.end()) started = f->second;
    }
    const auto t0 = std::chrono::steady_clock::now();
    CodeObject code;
    if (started.valid()) {
        const Compiled &done = started.get();   // (waits for it)
        code = done.code;
        if (!code) log = done.log;
        std::lock_guard<std::mutex> hold(g_cache_lock);
        g_in_flight.erase(memory_key);
    }
    if (!code) code = compile_now(source, memory_key, log);
    seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return code;
}

bool split_parts(const std::string &source, std::vector<std::string> &parts);

}  // namespace

void precompile_scene_source(const std::string &source) {
    std::vector<std::string> parts;
    if (helper_path().empty()) return;   // in this process the compiler would only compete with the caller's own work
    if (!split_parts(source, parts)) parts.assign(1, source);
    for (const std::string &part : parts) {
        const std::string memory_key = memory_key_of(part);
        std::lock_guard<std::mutex> hold(g_cache_lock);
        if (g_cache.count(memory_key) || g_in_flight.count(memory_key)) continue;
        if (g_in_flight.size() >= 64) return;   // a caller that announces scenes and never uses them
        g_in_flight.emplace(memory_key, std::async(std::launch::async, [part, memory_key] {
                                            Compiled out;
                                            out.code = compile_now(part, memory_key, out.log);
                                            return out;
                                        }).share());
    }
}

namespace {

// The generated source's parts (codegen.cpp marks them: the lines "//@part frame", "//@part exact" and "//@part eval" switch
// between them; what stands before the first mark belongs to all): the frame's kernel, the exact pass behind it, the
// evaluation at given points.  Parts without a kernel are left out; fewer than two parts: false, one program serves.
// The exact pass runs over the tiles the frame's kernel marks, as good as never, and is the part that takes longest to
// compile: it is compiled at -O1 ($RM_EXACT_OPT names another level), a third faster, so that it does not hold the
// scene's first frame up.
bool split_parts(const std::string &source, std::vector<std::string> &parts) {
    parts.clear();
    if (source.find("//@part frame\n") == std::string::npos) return false;
    std::string text[3];
    bool kernel[3] = { false, false, false };
    int part = -1;   // -1: all
    for (size_t at = 0; at < source.size();) {
        size_t end = source.find('\n', at);
        end = end == std::string::npos ? source.size() : end + 1;
        const std::string line = source.substr(at, end - at);
        at = end;
        if (line == "//@part frame\n") { part = 0; continue; }
        if (line == "//@part exact\n") { part = 1; continue; }
        if (line == "//@part eval\n") { part = 2; continue; }
        for (int q = 0; q < 3; q++)
            if (part < 0 || part == q) text[q] += line;
        if (part >= 0 && line.find("__global__") != std::string::npos) kernel[part] = true;
    }
    const char *level = std::getenv("RM_EXACT_OPT");
    if (!level || !*level) level = "-O1";
    for (int q = 0; q < 3; q++) {
        if (!kernel[q]) continue;
        parts.push_back(q == 1 && std::strcmp(level, "-O3") != 0 ? "//@optimise " + std::string(level) + "\n" + text[q] : text[q]);
    }
    return parts.size() >= 2;
}

}  // namespace

bool compile_scene_module(const std::string &source, uint32_t points_per_lane, SceneModule &out, std::string &log) {
    out.unload();
    // With the compiler in processes of its own the parts compile side by side (inside one process the compiler library
    // serialises its programs: nothing to gain there).
    std::vector<std::string> parts;
    if (helper_path().empty() || !split_parts(source, parts)) parts.assign(1, source);
    std::vector<CodeObject> codes(parts.size());
    std::vector<std::string> logs(parts.size());
    std::vector<double> took(parts.size(), 0.0);
    double seconds = 0.0;
    {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> others;
        for (size_t q = 1; q < parts.size(); q++) others.emplace_back([&, q] { codes[q] = obtain_code(parts[q], logs[q], took[q]); });
        codes[0] = obtain_code(parts[0], logs[0], took[0]);
        for (std::thread &t : others) t.join();
        bool compiled = false;
        for (size_t q = 0; q < parts.size(); q++) {
            if (!codes[q]) {
                log = logs[q];
                return false;
            }
            compiled = compiled || took[q] > 0.0;
        }
        if (compiled) seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    hipModule_t loaded[3] = { nullptr, nullptr, nullptr };
    for (size_t q = 0; q < parts.size(); q++) {
        const hipError_t e = hipModuleLoadData(q == 0 ? &out.module : &out.more_modules[q - 1], codes[q]->data());
        if (e != hipSuccess) {
            log = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
            out.unload();
            return false;
        }
        loaded[q] = q == 0 ? out.module : out.more_modules[q - 1];
    }
    // the module among them that holds a kernel
    const auto find = [&](hipFunction_t *fn, const char *name) {
        hipError_t e = hipErrorNotFound;
        for (hipModule_t m : loaded) {
            if (!m) continue;
            e = hipModuleGetFunction(fn, m, name);
            if (e == hipSuccess) break;
        }
        (void)hipGetLastError();
        return e;
    };
    hipError_t e = hipSuccess;
    // a module holds the plain kernel or its work-counting variant (named apart, so that profiles tell them apart)
    const bool counting = source.find("rm_trace_counting") != std::string::npos;
    const struct { hipFunction_t *fn; const char *name; } symbols[] = { { &out.trace, counting ? "rm_trace_counting" : "rm_trace" },
                                                                         { &out.eval_sdf, "rm_eval_sdf" } };
    for (const auto &sym : symbols) {
        e = find(sym.fn, sym.name);
        if (e != hipSuccess) {
            log = std::string("hipModuleGetFunction(") + sym.name + "): " + hipGetErrorString(e);
            out.unload();
            return false;
        }
    }
    if (source.find("rm_trace_exact") != std::string::npos) {
        e = find(&out.trace_exact, "rm_trace_exact");
        if (e != hipSuccess) {
            log = std::string("hipModuleGetFunction(rm_trace_exact): ") + hipGetErrorString(e);
            out.unload();
            return false;
        }
    }
    if (source.find("rm_trace_refill") != std::string::npos) {
        const char *name = counting ? "rm_trace_refill_counting" : "rm_trace_refill";
        e = find(&out.trace_refill, name);
        if (e != hipSuccess) {
            log = std::string("hipModuleGetFunction(") + name + "): " + hipGetErrorString(e);
            out.unload();
            return false;
        }
    }
    out.points_per_lane = points_per_lane;
    out.compile_seconds = seconds;
    return true;
}

hipError_t launch_trace_module(const SceneModule &m, const SceneView &sc, const FrameView &fr, uint32_t waves_per_side, Sample *samples,
                               WorkCounters *counters, hipStream_t stream) {
    const uint32_t sw = fr.width * fr.kernel;
    const uint32_t sh = fr.band_count * fr.band_rows * fr.kernel;
    if (sw == 0 || sh == 0) return hipSuccess;
    const uint32_t pts = m.points_per_lane;
    SceneView sc_arg = sc;
    FrameView fr_arg = fr;
    void *args[] = { &sc_arg, &fr_arg, &samples, &counters };
    if (fr.refill_w_log2 != 0u) {
        // one wave per tile of 2^w x 2^h samples; LDS: the lattice tables and the queue of points awaiting their colour
        const uint32_t tiles_x = (sw + (1u << fr.refill_w_log2) - 1u) >> fr.refill_w_log2;
        const uint32_t tiles_y = (sh + (1u << fr.refill_h_log2) - 1u) >> fr.refill_h_log2;
        return hipModuleLaunchKernel(m.trace_refill, tiles_x, tiles_y, 1, 64u, 1, 1, (sc.lattice_words + 4u * 128u) * 4u, stream, args, nullptr);
    }
    const uint32_t side = 8u * waves_per_side;
    const uint32_t tiles_x = (sw + side * pts - 1u) / (side * pts);
    const uint32_t extra_rows = fr.tile_order ? (fr.extra_slots + tiles_x - 1u) / tiles_x : 0u;
    // generated kernels keep nested lists in registers: no parking area behind the lattice tables
    const bool writes_pixels = fr.pixels_f32 || fr.pixels_rgba8;
    const uint32_t lds = (sc.lattice_words + (writes_pixels ? 3u * side * side : 0u)) * 4u;
    return hipModuleLaunchKernel(m.trace, tiles_x, (sh + side - 1u) / side + extra_rows, 1, 64u * waves_per_side * waves_per_side, 1, 1,
                                 lds, stream, args, nullptr);
}

bool exact_launch_can_resolve(uint32_t kernel) { return kernel * kernel <= 64u; }

hipError_t launch_trace_exact_module(const SceneModule &m, const SceneView &sc, const FrameView &fr, uint32_t waves_per_side, Sample *samples,
                                     float *out_f32, unsigned char *out_rgba8, hipStream_t stream) {
    const uint32_t sw = fr.width * fr.kernel;
    const uint32_t sh = fr.band_count * fr.band_rows * fr.kernel;
    if (sw == 0 || sh == 0 || !m.trace_exact || !fr.tile_flags) return hipSuccess;
    const uint32_t side = 8u * waves_per_side, pts = m.points_per_lane;
    const uint32_t tiles_x = (sw + side * pts - 1u) / (side * pts), tiles_y = (sh + side - 1u) / side;
    const bool resolving = out_f32 || out_rgba8;
    if (resolving && !exact_launch_can_resolve(fr.kernel)) return hipErrorInvalidValue;   // (the caller asks first)
    SceneView sc_arg = sc;
    FrameView fr_arg = fr;
    fr_arg.tile_order = nullptr;   // whole tiles in natural order, nothing timed
    fr_arg.extra_slots = 0u;
    fr_arg.tile_cost = nullptr;
    fr_arg.slot_stats = nullptr;
    void *args[] = { &sc_arg, &fr_arg, &samples, &out_f32, &out_rgba8 };
    const bool writes_pixels = fr.pixels_f32 || fr.pixels_rgba8;
    const uint32_t lds = (sc.lattice_words + (writes_pixels ? 3u * side * side : 0u)) * 4u;
    // 64 tiles per workgroup; resolving: the k x k tiles behind a workgroup's (tile side)^2 pixels
    const uint32_t gx = resolving ? (tiles_x + fr.kernel - 1u) / fr.kernel : (uint32_t)(((size_t)tiles_x * tiles_y + 63u) / 64u);
    const uint32_t gy = resolving ? (tiles_y + fr.kernel - 1u) / fr.kernel : 1u;
    return hipModuleLaunchKernel(m.trace_exact, gx, gy, 1, 64u * waves_per_side * waves_per_side, 1, 1, lds, stream, args, nullptr);
}

hipError_t launch_eval_sdf_module(const SceneModule &m, const SceneView &sc, const float *xyz, uint32_t n, float *sd,
                                  uint32_t *prim, float *sd_march, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    SceneView sc_arg = sc;
    void *args[] = { &sc_arg, &xyz, &n, &sd, &prim, &sd_march };
    return hipModuleLaunchKernel(m.eval_sdf, (n + 255u) / 256u, 1, 1, 256, 1, 1, sc.lattice_words * 4u, stream, args, nullptr);
}

}  // namespace rmdev
