// polymesh_io.cpp -- see polymesh_io.hpp.
#include "polymesh_io.hpp"

#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
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

namespace {
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
        char* e = nullptr;
        const long long v = std::strtoll(p, &e, 10);
        if (e == p) fail("expected an integer");
        p = e;
        return v;
    }
    double readDouble() {
        skipWs();
        char* e = nullptr;
        const double v = std::strtod(p, &e);
        if (e == p) fail("expected a number");
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

void readLabels(Scanner& s, const Header& h, std::vector<int32_t>& out) {
    const long long n = s.readInt();
    if (n < 0) s.fail("negative list size");
    out.resize((size_t)n);
    const char c = s.peek();
    if (c == '{') {   // uniform list N{v}
        ++s.p;
        const long long v = s.readInt();
        s.expect('}');
        std::fill(out.begin(), out.end(), (int32_t)v);
        return;
    }
    if (n == 0 && c != '(') return;
    if (h.binary) {
        if (*s.p != '(') s.fail("expected '(' before binary data");
        ++s.p;
        const size_t bytes = (size_t)n * h.labelBytes;
        if ((size_t)(s.end - s.p) < bytes) s.fail("truncated binary label list");
        if (h.labelBytes == 4) std::memcpy(out.data(), s.p, bytes);
        else for (long long i = 0; i < n; ++i) { int64_t v; std::memcpy(&v, s.p + 8 * i, 8); out[(size_t)i] = (int32_t)v; }
        s.p += bytes;
        s.expect(')');
    } else {
        s.expect('(');
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

void readPoints(const std::string& file, std::vector<double>& pts) {
    const std::string buf = slurp(file);
    Scanner s{buf.data(), buf.data() + buf.size(), file};
    const Header h = readHeader(s);
    const long long n = s.readInt();
    if (n < 0) s.fail("negative point count");
    pts.resize((size_t)n * 3);
    if (h.binary) {
        s.skipWs();
        if (*s.p != '(') s.fail("expected '(' before binary data");
        ++s.p;
        const size_t bytes = (size_t)n * 3 * h.scalarBytes;
        if ((size_t)(s.end - s.p) < bytes) s.fail("truncated binary point list");
        if (h.scalarBytes == 8) std::memcpy(pts.data(), s.p, bytes);
        else for (long long i = 0; i < 3 * n; ++i) { float v; std::memcpy(&v, s.p + 4 * i, 4); pts[(size_t)i] = v; }
        s.p += bytes;
        s.expect(')');
    } else {
        s.expect('(');
        for (long long i = 0; i < n; ++i) {
            s.expect('(');
            pts[3 * i] = s.readDouble(); pts[3 * i + 1] = s.readDouble(); pts[3 * i + 2] = s.readDouble();
            s.expect(')');
        }
        s.expect(')');
    }
}

void readLabelList(const std::string& file, std::vector<int32_t>& out) {
    const std::string buf = slurp(file);
    Scanner s{buf.data(), buf.data() + buf.size(), file};
    const Header h = readHeader(s);
    readLabels(s, h, out);
}

static void readFaces(const std::string& file, std::vector<int32_t>& off, std::vector<int32_t>& val) {
    const std::string buf = slurp(file);
    Scanner s{buf.data(), buf.data() + buf.size(), file};
    const Header h = readHeader(s);
    if (h.binary || h.cls == "faceCompactList") {
        readLabels(s, h, off);     // offsets (nFaces + 1)
        readLabels(s, h, val);
        if (off.empty()) off.push_back(0);
        return;
    }
    const long long n = s.readInt();
    s.expect('(');
    off.assign(1, 0);
    off.reserve((size_t)n + 1);
    val.clear();
    val.reserve((size_t)n * 4);
    for (long long i = 0; i < n; ++i) {
        const long long k = s.readInt();
        s.expect('(');
        for (long long j = 0; j < k; ++j) val.push_back((int32_t)s.readInt());
        s.expect(')');
        off.push_back((int32_t)val.size());
    }
    s.expect(')');
}

static void readBoundary(const std::string& file, std::vector<PatchInfo>& patches) {
    const std::string buf = slurp(file);
    Scanner s{buf.data(), buf.data() + buf.size(), file};
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
    readPoints((pointsDir.empty() ? dir : pointsDir) + "/points", m.points);
    readFaces(dir + "/faces", m.faceOffsets, m.facePoints);
    readLabelList(dir + "/owner", m.owner);
    readLabelList(dir + "/neighbour", m.neighbour);
    while (!m.neighbour.empty() && m.neighbour.back() < 0) m.neighbour.pop_back();   // old-style padded neighbour list
    readBoundary(dir + "/boundary", m.patches);
    const int32_t nF = (int32_t)m.faceOffsets.size() - 1;
    if ((int32_t)m.owner.size() != nF) throw std::runtime_error(dir + ": owner size does not match the number of faces");
    if (m.neighbour.size() > m.owner.size()) throw std::runtime_error(dir + ": more neighbours than faces");
    int32_t nc = -1;
    for (int32_t c : m.owner) nc = std::max(nc, c);
    for (int32_t c : m.neighbour) nc = std::max(nc, c);
    m.nCells = nc + 1;
    const int32_t nP = m.nPoints();
    for (int32_t v : m.facePoints) if (v < 0 || v >= nP) throw std::runtime_error(dir + "/faces: point label out of range");
    int32_t expectStart = m.nInternalFaces();
    for (const auto& p : m.patches) {
        if (p.startFace != expectStart || p.nFaces < 0) throw std::runtime_error(dir + "/boundary: patch " + p.name + " is not contiguous");
        expectStart += p.nFaces;
    }
    if (expectStart != nF) throw std::runtime_error(dir + "/boundary: patches do not cover all boundary faces");
}

void writePoints(const std::string& dir, const std::string& location, int32_t nPoints, const double* pts, bool binary, int precision) {
    makeDirs(dir);
    FILE* f = openOut(dir + "/points");
    writeHeader(f, binary, "vectorField", location, "points", "");
    std::fprintf(f, "%d\n(", nPoints);
    if (binary) {
        std::fwrite(pts, sizeof(double), (size_t)nPoints * 3, f);
    } else {
        std::fputc('\n', f);
        for (int32_t i = 0; i < nPoints; ++i)
            std::fprintf(f, "(%.*g %.*g %.*g)\n", precision, pts[3 * i], precision, pts[3 * i + 1], precision, pts[3 * i + 2]);
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
    else { std::fputc('\n', f); for (int64_t i = 0; i < n; ++i) std::fprintf(f, "%d\n", v[i]); }
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
            for (int32_t i = 0; i < nF; ++i) {
                const int32_t b = m.faceOffsets[i], e = m.faceOffsets[i + 1];
                std::fprintf(f, "%d(", e - b);
                for (int32_t k = b; k < e; ++k) std::fprintf(f, k + 1 < e ? "%d " : "%d", m.facePoints[k]);
                std::fputs(")\n", f);
            }
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
