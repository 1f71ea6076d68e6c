// polymesh_io.cpp -- see polymesh_io.hpp.
#include "polymesh_io.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <vector>

namespace smhost {

bool fileExists(const std::string& path) {
    struct stat st;
    return ::stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
bool dirExists(const std::string& path) {
    struct stat st;
    return ::stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
void makeDirs(const std::string& path) {
    std::string cur;
    for (size_t i = 0; i <= path.size(); ++i) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty() && !dirExists(cur) && ::mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST)
                throw std::runtime_error("cannot create directory " + cur);
        }
        if (i < path.size()) cur.push_back(path[i]);
    }
}

static bool gWriteCompression = false;
void setWriteCompression(bool on) { gWriteCompression = on; }

// <file>.gz (OpenFOAM's writeCompression on) is read when <file> itself does not exist, as OpenFOAM does
static std::string slurpGz(const std::string& file) {
    gzFile g = gzopen(file.c_str(), "rb");
    if (!g) throw std::runtime_error("cannot open " + file);
    std::string s;
    std::vector<char> buf(1 << 20);
    for (;;) {
        const int n = gzread(g, buf.data(), (unsigned)buf.size());
        if (n < 0) { gzclose(g); throw std::runtime_error("gzip read error on " + file); }
        if (n == 0) break;
        s.append(buf.data(), (size_t)n);
    }
    gzclose(g);
    return s;
}

static std::string slurp(const std::string& file) {
    FILE* f = std::fopen(file.c_str(), "rb");
    if (!f) {
        if (fileExists(file + ".gz")) return slurpGz(file + ".gz");
        throw std::runtime_error("cannot open " + file);
    }
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::string s;
    s.resize((size_t)n);
    if (n > 0 && std::fread(&s[0], 1, (size_t)n, f) != (size_t)n) { std::fclose(f); throw std::runtime_error("short read on " + file); }
    std::fclose(f);
    return s;
}

// ---- host threads of the list readers / writers ------------------------------------------------------------------------------
// The big lists of a polyMesh directory (points, faces, owner, neighbour) are cut into contiguous pieces, one per thread, each
// piece parsed / formatted on its own and the results joined in order: exactly what the serial loop reads or writes (the tests
// compare byte for byte).  SMHOST_IO_THREADS caps the thread count (default: the hardware's, at most 64; 1 = the serial loops).
static unsigned ioThreads() {
    static const unsigned n = [] {
        if (const char* e = std::getenv("SMHOST_IO_THREADS")) return std::max(1u, std::min((unsigned)std::atoi(e), 256u));
        return std::max(1u, std::min(std::thread::hardware_concurrency(), 64u));
    }();
    return n;
}
static int64_t ioGrain() {     // smallest piece worth a thread (bytes or records); tests lower it to cut small files too
    if (const char* e = std::getenv("SMHOST_IO_GRAIN")) return std::max<int64_t>(1, std::atoll(e));
    return 1 << 16;
}
static int ioParts(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)ioThreads(), n / ioGrain())); }

// fresh pages of the big lists as transparent huge pages (THP mode "madvise" on the GPU boxes): first touch of 4 KB pages runs at
// ~6 GB/s per thread and ~20 GB/s on all threads together (the kernel's fault path), of 2 MB pages at 17 and > 100 GB/s
static void adviseHuge(void* p, size_t bytes) {
    const uintptr_t huge = (uintptr_t)2 << 20;
    if (!p || bytes < ((size_t)8 << 20)) return;
    const uintptr_t a = ((uintptr_t)p + huge - 1) & ~(huge - 1), e = ((uintptr_t)p + bytes) & ~(huge - 1);
    if (e > a) (void)::madvise((void*)a, (size_t)(e - a), MADV_HUGEPAGE);
}
template <class V>
static void reserveHuge(V& v, size_t n) {      // (for vectors that are still empty)
    if (v.capacity() >= n) return;
    v.reserve(n);
    adviseHuge((void*)v.data(), v.capacity() * sizeof(typename V::value_type));
}

template <class F>
static void parallelParts(int parts, F&& f) {      // f(part) on its own thread; the first exception is re-thrown
    if (parts <= 1) { f(0); return; }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err((size_t)parts);
    th.reserve((size_t)parts);
    for (int t = 0; t < parts; ++t) th.emplace_back([&f, &err, t] { try { f(t); } catch (...) { err[(size_t)t] = std::current_exception(); } });
    for (auto& x : th) x.join();
    for (auto& e : err) if (e) std::rethrow_exception(e);
}
static void parallelCopy(void* dst, const void* src, size_t bytes) {     // (first touch of the destination and page faults of a mapped source on all threads)
    const int parts = ioParts((int64_t)bytes);
    parallelParts(parts, [&](int t) {
        const size_t b = bytes * (size_t)t / (size_t)parts, e = bytes * (size_t)(t + 1) / (size_t)parts;
        std::memcpy((char*)dst + b, (const char*)src + b, e - b);
    });
}

// A file's bytes: mapped (the readers never need a terminating NUL: every number is parsed within [p, end)), or -- <file>.gz
// where <file> is missing -- inflated into memory
struct FileBuf {
    const char* data = nullptr;
    size_t size = 0;
    void* map = nullptr;
    std::string owned;
    explicit FileBuf(const std::string& file) {
        const int fd = ::open(file.c_str(), O_RDONLY);
        if (fd < 0) {
            if (fileExists(file + ".gz")) { owned = slurpGz(file + ".gz"); data = owned.data(); size = owned.size(); return; }
            throw std::runtime_error("cannot open " + file);
        }
        struct stat st;
        if (::fstat(fd, &st) != 0) { ::close(fd); throw std::runtime_error("cannot stat " + file); }
        size = (size_t)st.st_size;
        if (size > 0) {
            map = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (map == MAP_FAILED) {      // (file systems without mmap: plain read)
                map = nullptr;
                ::close(fd);
                owned = slurp(file);
                data = owned.data(); size = owned.size();
                return;
            }
            (void)::madvise(map, size, MADV_WILLNEED);
            data = (const char*)map;
        }
        ::close(fd);
    }
    ~FileBuf() { if (map) ::munmap(map, size); }
    FileBuf(const FileBuf&) = delete;
    FileBuf& operator=(const FileBuf&) = delete;
};

namespace {
// ---- numbers within [p, end): no terminating NUL is needed (mapped files) ---------------------------------------------------
inline const char* parseInt(const char* p, const char* end, long long& out) {
    const char* q = p;
    bool neg = false;
    if (q < end && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
    if (q >= end || *q < '0' || *q > '9') return nullptr;
    unsigned long long v = 0;
    while (q < end && *q >= '0' && *q <= '9') v = v * 10 + (unsigned)(*q++ - '0');
    out = neg ? -(long long)v : (long long)v;
    return q;
}
// the token at p through strtod (any form strtod takes; a private NUL-terminated copy)
const char* parseDoubleSlow(const char* p, const char* end, double& out) {
    const char* q = p;
    while (q < end && !std::strchr(" \t\n\r;(){}", *q)) ++q;
    const std::string tok(p, q);
    char* e = nullptr;
    out = std::strtod(tok.c_str(), &e);
    if (e == tok.c_str()) return nullptr;
    return p + (e - tok.c_str());
}
// Decimal text -> double, correctly rounded like strtod: a mantissa of at most 19 digits that fits 2^53 times / over an exactly
// representable power of ten (|exponent| <= 22) is ONE correctly rounded IEEE operation on exact operands (Clinger's fast path);
// everything else (17-digit mantissas beyond 2^53, huge exponents, nan / inf, hex floats) goes through strtod itself.
inline const char* parseDouble(const char* p, const char* end, double& out) {
    static const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char* q = p;
    bool neg = false;
    if (q < end && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
    unsigned long long w = 0;
    int digits = 0, sig = 0, e10 = 0;
    while (q < end && *q >= '0' && *q <= '9') {
        if (sig > 0 || *q != '0') { if (sig < 19) { w = w * 10 + (unsigned)(*q - '0'); ++sig; } else return parseDoubleSlow(p, end, out); }
        ++digits; ++q;
    }
    if (q < end && *q == '.') {
        ++q;
        while (q < end && *q >= '0' && *q <= '9') {
            if (sig > 0 || *q != '0') { if (sig < 19) { w = w * 10 + (unsigned)(*q - '0'); ++sig; } else return parseDoubleSlow(p, end, out); }
            --e10; ++digits; ++q;
        }
    }
    if (digits == 0) return parseDoubleSlow(p, end, out);      // nan, inf, or not a number at all
    if (q < end && (*q == 'e' || *q == 'E')) {
        const char* r = q + 1;
        bool eneg = false;
        if (r < end && (*r == '-' || *r == '+')) { eneg = *r == '-'; ++r; }
        if (r < end && *r >= '0' && *r <= '9') {
            int ex = 0;
            while (r < end && *r >= '0' && *r <= '9') { if (ex < 100000) ex = ex * 10 + (*r - '0'); ++r; }
            e10 += eneg ? -ex : ex;
            q = r;
        }
    }
    if (q < end && (*q == 'x' || *q == 'X' || *q == 'p' || *q == 'P')) return parseDoubleSlow(p, end, out);     // (hex float)
    if (w <= (1ull << 53) && e10 >= -22 && e10 <= 22) {
        double v = (double)w;
        if (e10 < 0) v /= kPow10[-e10]; else v *= kPow10[e10];
        out = neg ? -v : v;
        return q;
    }
    if (w == 0) { out = neg ? -0.0 : 0.0; return q; }
#if defined(__x86_64__) && defined(__LDBL_MANT_DIG__) && (__LDBL_MANT_DIG__ == 64)
    // 17 .. 19 digit mantissas (coordinates written with precision 17): w < 2^64 and 10^|e10| (<= 10^27 < 2^90, an odd factor
    // 5^27 < 2^63) are exact in the x87 extended format, so ONE extended operation gives the quotient / product rounded to 64
    // mantissa bits, at most half a unit of that format from the exact value.  Rounding it once more to 53 bits is the correctly
    // rounded double unless the extended result sits within one unit of a midpoint between two doubles (low 11 bits 0x3ff ..
    // 0x401) -- then, and only then, strtod decides.  (The magnitudes, 1e-27 .. 1.9e46, are far from the ends of the double range.)
    if (e10 >= -27 && e10 <= 27) {
        static const long double kPow10L[28] = {1e0L, 1e1L, 1e2L, 1e3L, 1e4L, 1e5L, 1e6L, 1e7L, 1e8L, 1e9L, 1e10L, 1e11L, 1e12L, 1e13L, 1e14L,
                                                1e15L, 1e16L, 1e17L, 1e18L, 1e19L, 1e20L, 1e21L, 1e22L, 1e23L, 1e24L, 1e25L, 1e26L, 1e27L};
        long double v = (long double)w;
        if (e10 < 0) v /= kPow10L[-e10]; else v *= kPow10L[e10];
        unsigned long long mant;
        std::memcpy(&mant, &v, 8);      // (the 64 explicit mantissa bits of the extended format)
        const unsigned low = (unsigned)(mant & 0x7ffull);
        if (low < 0x3ffu || low > 0x401u) {
            const double dv = (double)v;
            out = neg ? -dv : dv;
            return q;
        }
    }
#endif
    return parseDoubleSlow(p, end, out);
}

struct Scanner {
    const char* p;
    const char* end;
    std::string file;
    [[noreturn]] void fail(const std::string& what) const { throw std::runtime_error(file + ": " + what); }
    void skipWs() {
        while (p < end) {
            if (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') { ++p; continue; }
            if (*p == '/' && p + 1 < end && p[1] == '/') { while (p < end && *p != '\n') ++p; continue; }
            if (*p == '/' && p + 1 < end && p[1] == '*') {
                p += 2;
                while (p + 1 < end && !(p[0] == '*' && p[1] == '/')) ++p;
                p = (p + 1 < end) ? p + 2 : end;
                continue;
            }
            break;
        }
    }
    bool eof() { skipWs(); return p >= end; }
    char peek() { skipWs(); return p < end ? *p : '\0'; }
    void expect(char c) {
        skipWs();
        if (p >= end || *p != c) fail(std::string("expected '") + c + "'");
        ++p;
    }
    long long readInt() {
        skipWs();
        long long v;
        const char* e = parseInt(p, end, v);
        if (!e) fail("expected an integer");
        p = e;
        return v;
    }
    double readDouble() {
        skipWs();
        double v;
        const char* e = parseDouble(p, end, v);
        if (!e) fail("expected a number");
        p = e;
        return v;
    }
    std::string readWord() {
        skipWs();
        if (p < end && *p == '"') {
            const char* b = ++p;
            while (p < end && *p != '"') ++p;
            std::string s(b, p);
            if (p < end) ++p;
            return s;
        }
        const char* b = p;
        while (p < end && !std::strchr(" \t\n\r;(){}", *p)) ++p;
        return std::string(b, p);
    }
    // value of a dictionary entry: everything up to the terminating ';' (nested brackets kept)
    std::string readValue() {
        skipWs();
        std::string v;
        int depth = 0;
        while (p < end) {
            if (*p == '"') { const char* b = p++; while (p < end && *p != '"') ++p; if (p < end) ++p; v.append(b, p); continue; }
            if (*p == '(' || *p == '{') ++depth;
            if (*p == ')' || *p == '}') --depth;
            if (*p == ';' && depth <= 0) { ++p; break; }
            v.push_back(*p++);
        }
        while (!v.empty() && std::isspace((unsigned char)v.back())) v.pop_back();
        return v;
    }
    std::map<std::string, std::string> readDict() {   // after '{'
        std::map<std::string, std::string> d;
        while (true) {
            const char c = peek();
            if (c == '}') { ++p; break; }
            if (c == '\0') fail("unterminated dictionary");
            const std::string key = readWord();
            if (key.empty()) fail("bad dictionary entry");
            if (peek() == '{') { ++p; readDict(); continue; }   // nested sub-dictionary: skipped
            d[key] = readValue();
        }
        return d;
    }
};

struct Header {
    bool binary = false;
    int labelBytes = 4, scalarBytes = 8;
    std::string cls;
};

Header readHeader(Scanner& s) {
    Header h;
    s.skipWs();
    const std::string w = s.readWord();
    if (w != "FoamFile") s.fail("missing FoamFile header");
    s.expect('{');
    auto d = s.readDict();
    auto unq = [](std::string v) { if (v.size() >= 2 && v.front() == '"') v = v.substr(1, v.size() - 2); return v; };
    if (d.count("format")) h.binary = (d["format"] == "binary");
    if (d.count("class")) h.cls = d["class"];
    if (d.count("arch")) {
        const std::string a = unq(d["arch"]);
        if (a.find("label=64") != std::string::npos) h.labelBytes = 8;
        if (a.find("scalar=32") != std::string::npos) h.scalarBytes = 4;
    }
    return h;
}

// ---- ascii list bodies on several threads ------------------------------------------------------------------------------------
// [b, e) = the body of a list between its outer brackets, cut into `parts` pieces; a cut is moved forward to just behind the next
// character of `delims` (a record never straddles two pieces).  Bodies with comments (never written by OpenFOAM inside a list)
// or anything else a piece cannot take make parsePiece return false: the caller then reads the list with the serial Scanner,
// which also words the errors.
inline bool plainWs(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
std::vector<const char*> cutBody(const char* b, const char* e, int parts, const char* delims) {
    std::vector<const char*> cut((size_t)parts + 1);
    cut[0] = b;
    cut[(size_t)parts] = e;
    for (int t = 1; t < parts; ++t) {
        const char* q = b + (size_t)(e - b) * (size_t)t / (size_t)parts;
        q = std::max(q, cut[(size_t)t - 1]);
        while (q < e && !std::strchr(delims, *q)) ++q;
        cut[(size_t)t] = q < e ? q + 1 : e;
    }
    return cut;
}
// the outer closing bracket of the list that starts at `open` ('('): the last ')' of the file that is followed by nothing but white
// space and comments without brackets (the footer line); nullptr when the file does not look like that
const char* lastCloser(const char* open, const char* end) {
    const char* q = end;
    while (q > open && q[-1] != ')') { if (q[-1] == '(') return nullptr; --q; }
    return q > open + 0 && q[-1] == ')' ? q - 1 : nullptr;
}

// What one thread parses out of its piece of a list: a plain uninitialised buffer written through a LOCAL pointer (a
// std::vector reached through a reference keeps its end pointer in memory: every push_back loads and stores it -- the parse ran
// at 150 MB/s per thread), sized from the piece's length (no number is shorter than two characters with its separator)
template <class T>
struct RawBuf {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    RawBuf() = default;
    RawBuf(const RawBuf&) = delete;
    RawBuf& operator=(const RawBuf&) = delete;
    ~RawBuf() { std::free(p); }
    void alloc(size_t c) {
        cap = std::max<size_t>(c, 16);
        p = (T*)std::malloc(cap * sizeof(T));
        if (!p) throw std::bad_alloc();
        adviseHuge(p, cap * sizeof(T));
    }
    size_t size() const { return n; }
};
template <class T>
void joinPieces(std::vector<T>& out, std::vector<RawBuf<T>>& pieces) {
    std::vector<size_t> base(pieces.size() + 1, 0);
    for (size_t i = 0; i < pieces.size(); ++i) base[i + 1] = base[i] + pieces[i].n;
    { std::vector<T> fresh; reserveHuge(fresh, base.back()); out.swap(fresh); }
    out.resize(base.back());
    parallelParts((int)pieces.size(), [&](int t) {
        RawBuf<T>& b = pieces[(size_t)t];
        if (b.n) std::memcpy(out.data() + base[(size_t)t], b.p, b.n * sizeof(T));
        std::free(b.p); b.p = nullptr;
    });
}

// n labels between s.p ('(' already taken) and the closing bracket; false: not taken (s.p untouched)
bool readLabelBodyParallel(Scanner& s, long long n, std::vector<int32_t>& out) {
    const char* close = lastCloser(s.p, s.end);
    if (!close) return false;
    const int parts = ioParts((int64_t)(close - s.p));
    if (parts <= 1) return false;
    // (a label list may be followed by a second list in the same file -- faceCompactList -- whose brackets lastCloser would see:
    // the caller only comes here for the LAST list of a file)
    const auto cut = cutBody(s.p, close, parts, " \n\t\r");
    std::vector<RawBuf<int32_t>> pieces((size_t)parts);
    std::atomic<bool> ok{true};
    parallelParts(parts, [&](int t) {
        const char* q = cut[(size_t)t];
        const char* e = cut[(size_t)t + 1];
        pieces[(size_t)t].alloc((size_t)(e - q) / 2 + 16);
        int32_t* o = pieces[(size_t)t].p;
        while (true) {
            while (q < e && plainWs(*q)) ++q;
            if (q >= e) break;
            long long x;
            const char* r = parseInt(q, e, x);
            if (!r || (r < e && !plainWs(*r))) { ok = false; return; }
            *o++ = (int32_t)x;
            q = r;
        }
        pieces[(size_t)t].n = (size_t)(o - pieces[(size_t)t].p);
    });
    if (!ok) return false;
    size_t total = 0;
    for (auto& v : pieces) total += v.size();
    if ((long long)total != n) return false;
    joinPieces(out, pieces);
    s.p = close + 1;
    return true;
}

// `last`: this is the file's last list (the several-thread ascii reader looks for the closing bracket from the file's end)
void readLabels(Scanner& s, const Header& h, std::vector<int32_t>& out, bool last = true) {
    const long long n = s.readInt();
    if (n < 0) s.fail("negative list size");
    if (n > INT32_MAX) s.fail("list size beyond 32-bit labels");
    const char c = s.peek();
    if (c == '{') {   // uniform list N{v}
        ++s.p;
        const long long v = s.readInt();
        s.expect('}');
        out.assign((size_t)n, (int32_t)v);
        return;
    }
    if (n == 0 && c != '(') { out.clear(); return; }
    if (n > s.end - s.p) s.fail("list size exceeds the file");     // (every element takes at least a byte: nothing is allocated for a bad count)
    if (h.binary) {
        if (c != '(') s.fail("expected '(' before binary data");
        ++s.p;
        const size_t bytes = (size_t)n * h.labelBytes;
        if ((size_t)(s.end - s.p) < bytes) s.fail("truncated binary label list");
        { std::vector<int32_t> fresh; reserveHuge(fresh, (size_t)n); out.swap(fresh); }
        out.resize((size_t)n);
        if (h.labelBytes == 4) parallelCopy(out.data(), s.p, bytes);
        else {
            const int parts = ioParts(n);
            const char* src = s.p;
            parallelParts(parts, [&](int t) {
                for (long long i = n * t / parts, e = n * (t + 1) / parts; i < e; ++i) { int64_t v; std::memcpy(&v, src + 8 * i, 8); out[(size_t)i] = (int32_t)v; }
            });
        }
        s.p += bytes;
        s.expect(')');
    } else {
        s.expect('(');
        if (last && readLabelBodyParallel(s, n, out)) return;
        out.resize((size_t)n);
        for (long long i = 0; i < n; ++i) out[(size_t)i] = (int32_t)s.readInt();
        s.expect(')');
    }
}

const char* kBanner =
    "/*--------------------------------*- C++ -*----------------------------------*\\\n"
    "  =========                 |\n"
    "  \\\\      /  F ield         | smoothMesh (MI355X engine): polyMesh written by smoothmesh_amd\n"
    "   \\\\    /   O peration     |\n"
    "    \\\\  /    A nd           |\n"
    "     \\\\/     M anipulation  |\n"
    "\\*---------------------------------------------------------------------------*/\n";

void writeHeader(FILE* f, bool binary, const std::string& cls, const std::string& location, const std::string& object,
                 const std::string& note) {
    std::fputs(kBanner, f);
    std::fprintf(f, "FoamFile\n{\n    version     2.0;\n    format      %s;\n", binary ? "binary" : "ascii");
    std::fputs("    arch        \"LSB;label=32;scalar=64\";\n", f);
    if (!note.empty()) std::fprintf(f, "    note        \"%s\";\n", note.c_str());
    std::fprintf(f, "    class       %s;\n    location    \"%s\";\n    object      %s;\n}\n", cls.c_str(), location.c_str(), object.c_str());
    std::fputs("// * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * //\n\n\n", f);
}
void writeFooter(FILE* f) { std::fputs("\n\n// ************************************************************************* //\n", f); }

FILE* openOut(const std::string& file) {
    FILE* f = std::fopen(file.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + file);
    return f;
}
// close; with write compression the file becomes <file>.gz.  The other variant is removed either way so that a
// reader never finds a stale copy.
void closeOut(FILE* f, const std::string& file) {
    if (std::fclose(f) != 0) throw std::runtime_error("write error on " + file);
    if (!gWriteCompression) { std::remove((file + ".gz").c_str()); return; }
    const std::string data = slurp(file);
    gzFile g = gzopen((file + ".gz").c_str(), "wb");
    if (!g) throw std::runtime_error("cannot write " + file + ".gz");
    size_t off = 0;
    while (off < data.size()) {
        const unsigned chunk = (unsigned)std::min<size_t>(data.size() - off, 1u << 30);
        if (gzwrite(g, data.data() + off, chunk) != (int)chunk) { gzclose(g); throw std::runtime_error("gzip write error on " + file + ".gz"); }
        off += chunk;
    }
    if (gzclose(g) != Z_OK) throw std::runtime_error("gzip write error on " + file + ".gz");
    std::remove(file.c_str());
}
}  // namespace

// n "(x y z)" records between s.p (the outer '(' already taken) and the closing bracket, on several threads; false: not taken
static bool readPointBodyParallel(Scanner& s, long long n, std::vector<double>& pts) {
    const char* close = lastCloser(s.p, s.end);
    if (!close) return false;
    const int parts = ioParts((int64_t)(close - s.p));
    if (parts <= 1) return false;
    const auto cut = cutBody(s.p, close, parts, ")");
    std::vector<RawBuf<double>> pieces((size_t)parts);
    std::atomic<bool> ok{true};
    parallelParts(parts, [&](int t) {
        const char* q = cut[(size_t)t];
        const char* e = cut[(size_t)t + 1];
        pieces[(size_t)t].alloc((size_t)(e - q) / 2 + 16);
        double* o = pieces[(size_t)t].p;
        struct Done { RawBuf<double>& b; double*& o; ~Done() { b.n = (size_t)(o - b.p); } } done{pieces[(size_t)t], o};
        auto ws = [&] { while (q < e && plainWs(*q)) ++q; };
        while (true) {
            ws();
            if (q >= e) break;
            if (*q != '(') { ok = false; return; }
            ++q;
            for (int k = 0; k < 3; ++k) {
                ws();
                double x;
                const char* r = q < e ? parseDouble(q, e, x) : nullptr;
                if (!r) { ok = false; return; }
                *o++ = x;
                q = r;
            }
            ws();
            if (q >= e || *q != ')') { ok = false; return; }
            ++q;
        }
    });
    if (!ok) return false;
    size_t total = 0;
    for (auto& v : pieces) total += v.size();
    if ((long long)total != 3 * n) return false;
    joinPieces(pts, pieces);
    s.p = close + 1;
    return true;
}

void readPoints(const std::string& file, std::vector<double>& pts) {
    const FileBuf buf(file);
    Scanner s{buf.data, buf.data + buf.size, file};
    const Header h = readHeader(s);
    const long long n = s.readInt();
    if (n < 0) s.fail("negative point count");
    if (n > INT32_MAX) s.fail("point count beyond 32-bit labels");
    if (n > s.end - s.p) s.fail("point count exceeds the file");
    if (h.binary) {
        s.skipWs();
        if (s.p >= s.end || *s.p != '(') s.fail("expected '(' before binary data");
        ++s.p;
        const size_t bytes = (size_t)n * 3 * h.scalarBytes;
        if ((size_t)(s.end - s.p) < bytes) s.fail("truncated binary point list");
        { std::vector<double> fresh; reserveHuge(fresh, (size_t)n * 3); pts.swap(fresh); }
        pts.resize((size_t)n * 3);
        if (h.scalarBytes == 8) parallelCopy(pts.data(), s.p, bytes);
        else for (long long i = 0; i < 3 * n; ++i) { float v; std::memcpy(&v, s.p + 4 * i, 4); pts[(size_t)i] = v; }
        s.p += bytes;
        s.expect(')');
    } else {
        s.expect('(');
        if (readPointBodyParallel(s, n, pts)) return;
        pts.resize((size_t)n * 3);
        for (long long i = 0; i < n; ++i) {
            s.expect('(');
            pts[3 * i] = s.readDouble(); pts[3 * i + 1] = s.readDouble(); pts[3 * i + 2] = s.readDouble();
            s.expect(')');
        }
        s.expect(')');
    }
}

void readLabelList(const std::string& file, std::vector<int32_t>& out) {
    const FileBuf buf(file);
    Scanner s{buf.data, buf.data + buf.size, file};
    const Header h = readHeader(s);
    readLabels(s, h, out);
}

// n "k(a b ...)" records of an ascii faceList on several threads (s.p behind the outer '('); false: not taken
static bool readFaceBodyParallel(Scanner& s, long long n, std::vector<int32_t>& off, std::vector<int32_t>& val) {
    const char* close = lastCloser(s.p, s.end);
    if (!close) return false;
    const int parts = ioParts((int64_t)(close - s.p));
    if (parts <= 1) return false;
    const auto cut = cutBody(s.p, close, parts, ")");
    std::vector<RawBuf<int32_t>> sizes((size_t)parts), vals((size_t)parts);
    std::atomic<bool> ok{true};
    parallelParts(parts, [&](int t) {
        const char* q = cut[(size_t)t];
        const char* e = cut[(size_t)t + 1];
        sizes[(size_t)t].alloc((size_t)(e - q) / 3 + 16);      // ("0()" is the shortest face record)
        vals[(size_t)t].alloc((size_t)(e - q) / 2 + 16);
        int32_t* so = sizes[(size_t)t].p;
        int32_t* vo = vals[(size_t)t].p;
        struct Done { RawBuf<int32_t>& a; int32_t*& ao; RawBuf<int32_t>& b; int32_t*& bo; ~Done() { a.n = (size_t)(ao - a.p); b.n = (size_t)(bo - b.p); } }
            done{sizes[(size_t)t], so, vals[(size_t)t], vo};
        auto ws = [&] { while (q < e && plainWs(*q)) ++q; };
        while (true) {
            ws();
            if (q >= e) break;
            long long k;
            const char* r = parseInt(q, e, k);
            if (!r || k < 0) { ok = false; return; }
            q = r;
            ws();
            if (q >= e || *q != '(') { ok = false; return; }
            ++q;
            for (long long j = 0; j < k; ++j) {
                ws();
                long long x;
                r = q < e ? parseInt(q, e, x) : nullptr;
                if (!r) { ok = false; return; }
                *vo++ = (int32_t)x;
                q = r;
            }
            ws();
            if (q >= e || *q != ')') { ok = false; return; }
            ++q;
            *so++ = (int32_t)k;
        }
    });
    if (!ok) return false;
    std::vector<size_t> fBase((size_t)parts + 1, 0), vBase((size_t)parts + 1, 0);
    for (int t = 0; t < parts; ++t) { fBase[(size_t)t + 1] = fBase[(size_t)t] + sizes[(size_t)t].size(); vBase[(size_t)t + 1] = vBase[(size_t)t] + vals[(size_t)t].size(); }
    if ((long long)fBase.back() != n || vBase.back() > (size_t)INT32_MAX) return false;
    { std::vector<int32_t> fresh; reserveHuge(fresh, (size_t)n + 1); off.swap(fresh); }
    off.resize((size_t)n + 1);
    off[0] = 0;
    parallelParts(parts, [&](int t) {
        int32_t o = (int32_t)vBase[(size_t)t];
        int32_t* dst = off.data() + fBase[(size_t)t] + 1;
        const RawBuf<int32_t>& sz = sizes[(size_t)t];
        for (size_t i = 0; i < sz.n; ++i) { o += sz.p[i]; *dst++ = o; }
    });
    joinPieces(val, vals);
    s.p = close + 1;
    return true;
}

static void readFaces(const std::string& file, std::vector<int32_t>& off, std::vector<int32_t>& val) {
    const FileBuf buf(file);
    Scanner s{buf.data, buf.data + buf.size, file};
    const Header h = readHeader(s);
    if (h.binary || h.cls == "faceCompactList") {
        readLabels(s, h, off, false);     // offsets (nFaces + 1)
        readLabels(s, h, val, true);
        if (off.empty()) off.push_back(0);
        return;
    }
    const long long n = s.readInt();
    if (n < 0 || n >= INT32_MAX) s.fail("bad face count");
    if (n > s.end - s.p) s.fail("face count exceeds the file");
    s.expect('(');
    if (readFaceBodyParallel(s, n, off, val)) return;
    off.assign(1, 0);
    off.reserve((size_t)n + 1);
    val.clear();
    val.reserve((size_t)n * 4);
    for (long long i = 0; i < n; ++i) {
        const long long k = s.readInt();
        if (k < 0 || k > s.end - s.p) s.fail("bad face size");
        s.expect('(');
        for (long long j = 0; j < k; ++j) val.push_back((int32_t)s.readInt());
        s.expect(')');
        off.push_back((int32_t)val.size());
    }
    s.expect(')');
}

static void readBoundary(const std::string& file, std::vector<PatchInfo>& patches) {
    const FileBuf buf(file);
    Scanner s{buf.data, buf.data + buf.size, file};
    readHeader(s);
    const long long n = s.readInt();
    s.expect('(');
    patches.clear();
    for (long long i = 0; i < n; ++i) {
        PatchInfo p;
        p.name = s.readWord();
        s.expect('{');
        auto d = s.readDict();
        if (!d.count("type") || !d.count("nFaces") || !d.count("startFace")) s.fail("patch " + p.name + " lacks type/nFaces/startFace");
        p.type = d["type"];
        p.nFaces = (int32_t)std::atol(d["nFaces"].c_str());
        p.startFace = (int32_t)std::atol(d["startFace"].c_str());
        if (d.count("myProcNo")) p.myProcNo = (int32_t)std::atol(d["myProcNo"].c_str());
        if (d.count("neighbProcNo")) p.neighbProcNo = (int32_t)std::atol(d["neighbProcNo"].c_str());
        patches.push_back(p);
    }
    s.expect(')');
}

void readPolyMesh(const std::string& dir, const std::string& pointsDir, PolyMeshData& m) {
    const bool verbose = std::getenv("SMHOST_IO_VERBOSE") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!verbose) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[smhost] read %-10s %.3f s (%u threads)\n", what, std::chrono::duration<double>(t1 - t0).count(), ioThreads());
        t0 = t1;
    };
    readPoints((pointsDir.empty() ? dir : pointsDir) + "/points", m.points);
    lap("points");
    readFaces(dir + "/faces", m.faceOffsets, m.facePoints);
    lap("faces");
    readLabelList(dir + "/owner", m.owner);
    lap("owner");
    readLabelList(dir + "/neighbour", m.neighbour);
    lap("neighbour");
    while (!m.neighbour.empty() && m.neighbour.back() < 0) m.neighbour.pop_back();   // old-style padded neighbour list
    readBoundary(dir + "/boundary", m.patches);
    const int32_t nF = (int32_t)m.faceOffsets.size() - 1;
    if ((int32_t)m.owner.size() != nF) throw std::runtime_error(dir + ": owner size does not match the number of faces");
    if (m.neighbour.size() > m.owner.size()) throw std::runtime_error(dir + ": more neighbours than faces");
    auto maxOf = [](const std::vector<int32_t>& v) {
        const int parts = ioParts((int64_t)v.size());
        std::vector<int32_t> mx((size_t)parts, -1);
        parallelParts(parts, [&](int t) {
            int32_t m0 = -1;
            for (size_t i = v.size() * (size_t)t / (size_t)parts, e = v.size() * (size_t)(t + 1) / (size_t)parts; i < e; ++i) m0 = std::max(m0, v[i]);
            mx[(size_t)t] = m0;
        });
        return *std::max_element(mx.begin(), mx.end());
    };
    m.nCells = std::max(maxOf(m.owner), maxOf(m.neighbour)) + 1;
    const int32_t nP = m.nPoints();
    {
        const auto& v = m.facePoints;
        const int parts = ioParts((int64_t)v.size());
        std::atomic<bool> bad{false};
        parallelParts(parts, [&](int t) {
            bool b = false;
            for (size_t i = v.size() * (size_t)t / (size_t)parts, e = v.size() * (size_t)(t + 1) / (size_t)parts; i < e; ++i) b |= (v[i] < 0 || v[i] >= nP);
            if (b) bad = true;
        });
        if (bad) throw std::runtime_error(dir + "/faces: point label out of range");
    }
    int32_t expectStart = m.nInternalFaces();
    for (const auto& p : m.patches) {
        if (p.startFace != expectStart || p.nFaces < 0) throw std::runtime_error(dir + "/boundary: patch " + p.name + " is not contiguous");
        expectStart += p.nFaces;
    }
    if (expectStart != nF) throw std::runtime_error(dir + "/boundary: patches do not cover all boundary faces");
    lap("checks");
}

namespace {
// printf("%.*g", precision, v) without the format interpreter: std::to_chars(general, precision) is specified to give exactly
// those characters in the C locale (tests/test_polymesh_io.py compares the two writers byte for byte); non-finite values and
// anything to_chars refuses go through snprintf itself
inline char* putG(char* out, char* cap, double v, int precision) {
    if (std::isfinite(v)) {
        const auto r = std::to_chars(out, cap, v, std::chars_format::general, precision);
        if (r.ec == std::errc()) return r.ptr;
    }
    return out + std::snprintf(out, (size_t)(cap - out), "%.*g", precision, v);
}
inline char* putInt(char* out, char* cap, long long v) { return std::to_chars(out, cap, v).ptr; }

// n records, record i appended by fmt(i, out) (at most maxRec bytes each), formatted on several threads into per-thread buffers
// and written behind what f holds so far, in order
template <class Fmt>
void writeRecords(FILE* f, const std::string& file, int64_t n, size_t maxRec, Fmt fmt) {
    const int parts = ioParts(n);
    std::vector<std::vector<char>> bufs((size_t)parts);
    parallelParts(parts, [&](int t) {
        const int64_t b = n * t / parts, e = n * (t + 1) / parts;
        auto& buf = bufs[(size_t)t];
        buf.resize(std::max<size_t>((size_t)(e - b) * std::min<size_t>(maxRec, 48) + maxRec, 4096));
        size_t used = 0;
        for (int64_t i = b; i < e; ++i) {
            if (buf.size() - used < maxRec) buf.resize(buf.size() + buf.size() / 2 + maxRec);
            used = (size_t)(fmt(i, buf.data() + used, buf.data() + buf.size()) - buf.data());
        }
        buf.resize(used);
    });
    if (parts == 1) {
        if (!bufs[0].empty() && std::fwrite(bufs[0].data(), 1, bufs[0].size(), f) != bufs[0].size()) throw std::runtime_error("write error on " + file);
        return;
    }
    // every thread writes its own piece at its own offset
    if (std::fflush(f) != 0) throw std::runtime_error("write error on " + file);
    const off_t base = ::ftello(f);
    std::vector<off_t> at((size_t)parts + 1, base);
    for (int t = 0; t < parts; ++t) at[(size_t)t + 1] = at[(size_t)t] + (off_t)bufs[(size_t)t].size();
    const int fd = ::fileno(f);
    if (::ftruncate(fd, at.back()) != 0) throw std::runtime_error("cannot size " + file);
    parallelParts(parts, [&](int t) {
        const auto& buf = bufs[(size_t)t];
        size_t done = 0;
        while (done < buf.size()) {
            const ssize_t w = ::pwrite(fd, buf.data() + done, buf.size() - done, at[(size_t)t] + (off_t)done);
            if (w <= 0) throw std::runtime_error("write error on " + file);
            done += (size_t)w;
        }
    });
    if (::fseeko(f, at.back(), SEEK_SET) != 0) throw std::runtime_error("write error on " + file);
}
}  // namespace

void writePoints(const std::string& dir, const std::string& location, int32_t nPoints, const double* pts, bool binary, int precision) {
    makeDirs(dir);
    FILE* f = openOut(dir + "/points");
    writeHeader(f, binary, "vectorField", location, "points", "");
    std::fprintf(f, "%d\n(", nPoints);
    if (binary) {
        std::fwrite(pts, sizeof(double), (size_t)nPoints * 3, f);
    } else {
        std::fputc('\n', f);
        const int prec = precision < 0 ? 6 : std::max(precision, 1);      // (printf: "%.0g" means one digit, a negative precision the default)
        const size_t maxRec = 3 * ((size_t)prec + 32) + 8;
        writeRecords(f, dir + "/points", nPoints, maxRec, [&](int64_t i, char* o, char* cap) {
            *o++ = '(';
            o = putG(o, cap, pts[3 * i], prec); *o++ = ' ';
            o = putG(o, cap, pts[3 * i + 1], prec); *o++ = ' ';
            o = putG(o, cap, pts[3 * i + 2], prec); *o++ = ')'; *o++ = '\n';
            return o;
        });
    }
    std::fputs(")\n", f);
    writeFooter(f);
    closeOut(f, dir + "/points");
}

void writeLabelList(const std::string& file, const std::string& location, const std::string& object, const std::string& cls,
                    int64_t n, const int32_t* v, bool binary, const std::string& note) {
    FILE* f = openOut(file);
    writeHeader(f, binary, cls, location, object, note);
    std::fprintf(f, "%lld\n(", (long long)n);
    if (binary) std::fwrite(v, sizeof(int32_t), (size_t)n, f);
    else {
        std::fputc('\n', f);
        writeRecords(f, file, n, 16, [&](int64_t i, char* o, char* cap) { o = putInt(o, cap, v[i]); *o++ = '\n'; return o; });
    }
    std::fputs(")\n", f);
    writeFooter(f);
    closeOut(f, file);
}

namespace {
// one OBJ record: keyword + whitespace-separated tokens; "12/3/4" style vertex references keep their first number,
// negative numbers count back from the last vertex read
template <class OnVertex, class OnRecord>
void scanObj(const std::string& file, OnVertex onVertex, OnRecord onRecord) {
    std::ifstream in(file);
    if (!in) throw std::runtime_error("cannot open " + file);
    std::string line;
    int64_t nv = 0, lineNo = 0;
    while (std::getline(in, line)) {
        ++lineNo;
        std::istringstream ls(line);
        std::string key;
        if (!(ls >> key) || key[0] == '#') continue;
        if (key == "v") {
            double x, y, z;
            if (!(ls >> x >> y >> z)) throw std::runtime_error(file + ":" + std::to_string(lineNo) + ": bad vertex record");
            onVertex(x, y, z);
            ++nv;
        } else if (key == "f" || key == "l") {
            std::vector<int32_t> ids;
            std::string tok;
            while (ls >> tok) {
                const long i = std::strtol(tok.c_str(), nullptr, 10);
                const int64_t id = i > 0 ? i - 1 : nv + i;
                if (i == 0 || id < 0 || id >= nv) throw std::runtime_error(file + ":" + std::to_string(lineNo) + ": vertex reference out of range");
                ids.push_back((int32_t)id);
            }
            onRecord(key[0], ids);
        }
    }
}
}  // namespace

void readObjSurface(const std::string& file, std::vector<double>& points, std::vector<int32_t>& triangles) {
    points.clear(); triangles.clear();
    scanObj(file, [&](double x, double y, double z) { points.push_back(x); points.push_back(y); points.push_back(z); },
            [&](char kind, const std::vector<int32_t>& v) {
                if (kind != 'f') return;
                for (size_t k = 1; k + 1 < v.size(); ++k) { triangles.push_back(v[0]); triangles.push_back(v[k]); triangles.push_back(v[k + 1]); }
            });
}

void readObjEdges(const std::string& file, std::vector<double>& points, std::vector<int32_t>& edges) {
    std::vector<double> all;
    edges.clear();
    scanObj(file, [&](double x, double y, double z) { all.push_back(x); all.push_back(y); all.push_back(z); },
            [&](char kind, const std::vector<int32_t>& v) {
                if (kind != 'l') return;
                for (size_t k = 0; k + 1 < v.size(); ++k) { edges.push_back(v[k]); edges.push_back(v[k + 1]); }
            });
    std::vector<int32_t> renumber(all.size() / 3, -1);
    for (int32_t v : edges) renumber[(size_t)v] = 0;
    points.clear();
    int32_t next = 0;
    for (size_t i = 0; i < renumber.size(); ++i)
        if (renumber[i] == 0) {
            renumber[i] = next++;
            points.insert(points.end(), all.begin() + 3 * (std::ptrdiff_t)i, all.begin() + 3 * (std::ptrdiff_t)i + 3);
        }
    for (int32_t& v : edges) v = renumber[(size_t)v];
}

void writePolyMesh(const std::string& dir, const std::string& location, const PolyMeshData& m, bool binary, int precision) {
    makeDirs(dir);
    writePoints(dir, location, m.nPoints(), m.points.data(), binary, precision);
    const int32_t nF = m.nFaces();
    {
        FILE* f = openOut(dir + "/faces");
        writeHeader(f, binary, binary ? "faceCompactList" : "faceList", location, "faces", "");
        if (binary) {
            std::fprintf(f, "%d\n(", nF + 1);
            std::fwrite(m.faceOffsets.data(), sizeof(int32_t), (size_t)nF + 1, f);
            std::fprintf(f, ")\n\n%lld\n(", (long long)m.facePoints.size());
            std::fwrite(m.facePoints.data(), sizeof(int32_t), m.facePoints.size(), f);
            std::fputs(")\n", f);
        } else {
            std::fprintf(f, "%d\n(\n", nF);
            int32_t widest = 0;
            for (int32_t i = 0; i < nF; ++i) widest = std::max(widest, m.faceOffsets[i + 1] - m.faceOffsets[i]);
            writeRecords(f, dir + "/faces", nF, 16 + 12 * (size_t)widest, [&](int64_t i, char* o, char* cap) {
                const int32_t b = m.faceOffsets[i], e = m.faceOffsets[i + 1];
                o = putInt(o, cap, e - b); *o++ = '(';
                for (int32_t k = b; k < e; ++k) { o = putInt(o, cap, m.facePoints[k]); if (k + 1 < e) *o++ = ' '; }
                *o++ = ')'; *o++ = '\n';
                return o;
            });
            std::fputs(")\n", f);
        }
        writeFooter(f);
        closeOut(f, dir + "/faces");
    }
    char note[256];
    std::snprintf(note, sizeof note, "nPoints:%d  nCells:%d  nFaces:%d  nInternalFaces:%d", m.nPoints(), m.nCells, nF, m.nInternalFaces());
    writeLabelList(dir + "/owner", location, "owner", "labelList", nF, m.owner.data(), binary, note);
    writeLabelList(dir + "/neighbour", location, "neighbour", "labelList", m.nInternalFaces(), m.neighbour.data(), binary, note);
    {
        FILE* f = openOut(dir + "/boundary");
        writeHeader(f, false, "polyBoundaryMesh", location, "boundary", "");
        std::fprintf(f, "%d\n(\n", (int)m.patches.size());
        for (const auto& p : m.patches) {
            std::fprintf(f, "    %s\n    {\n        type            %s;\n", p.name.c_str(), p.type.c_str());
            if (p.type == "wall") std::fputs("        inGroups        1(wall);\n", f);
            std::fprintf(f, "        nFaces          %d;\n        startFace       %d;\n", p.nFaces, p.startFace);
            if (p.type == "processor")
                std::fprintf(f, "        matchTolerance  0.0001;\n        transform       unknown;\n        myProcNo        %d;\n        neighbProcNo    %d;\n",
                             p.myProcNo, p.neighbProcNo);
            std::fputs("    }\n", f);
        }
        std::fputs(")\n", f);
        writeFooter(f);
        closeOut(f, dir + "/boundary");
    }
}

}  // namespace smhost
