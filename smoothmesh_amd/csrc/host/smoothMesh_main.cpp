// smoothMesh_main.cpp -- the `smoothMesh` front-end: same command-line options, case-directory
// inputs/outputs and log lines as the reference's main() (src/smoothMesh.C:1635-2445), with the
// iteration loop (SM.C:2257-2437) executed by the MI355X engine behind include/smgpu.h.
//
// This host is standalone (own option parser, own polyMesh I/O) because OpenFOAM is not available
// in the build environment; INTEGRATION.md shows the OpenFOAM-linked variant of the same calls.
// Boundary layer treatment (-layerPatches ..., orthogonalBoundaryBlending.C) is available, serial and -parallel;
// boundary point smoothing (constant/geometry/*.obj, boundaryPointSmoothing.C), both serial and -parallel.
//
//   smoothMesh [-case <dir>] [-parallel] [-time <t|constant>] [-centroidalIters n] [-relTol x] ...
//
// -parallel: the case holds processorN/ sub-domains (decomposePar layout, with pointProcAddressing) and the run is what
// `mpirun -np N smoothMesh -parallel` is for the reference (testcase/run_parallel:19): ONE PROCESS PER SUB-DOMAIN, one GPU
// each.  There is no MPI in this image, so the front-end forks its N ranks itself before anything touches HIP
// (node_comm.hpp); rank r takes processor<r>/ and device (r + -device) mod #devices.  Per iteration the shared-point records
// (exchange A + the layer / boundary record L in ONE group, then exchange F; syncTools::syncPointList at SM.C:134-148,
// 402-478, 2374) travel as grouped ncclSend / ncclRecv between the ranks that share points, straight from and into the
// engines' device buffers on the engine's stream -- RCCL over xGMI, no host copy, no host synchronisation while relTol <= 0
// (the per-iteration {residual, nFrozenPoints} are then gathered once per chunk).  The small collectives of the set-up go
// through the ranks' shared mapping.  SMOOTHMESH_TRANSPORT=shm (chosen by itself when the node has fewer GPUs than ranks:
// RCCL refuses two ranks on one device) stages the same records through that mapping instead -- a debug transport.
#include <dirent.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <thread>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <regex>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/smgpu.h"
#include "node_comm.hpp"
#include "polymesh_io.hpp"

using namespace smhost;

namespace {

const double VSMALL = 1.0e-300, SMALL = 1.0e-15;

NodeComm g_comm;            // the ranks of a -parallel run (size 1 otherwise)
bool g_master = true;       // Info: only the master rank prints (Pstream::master())
#define OUT(...) do { if (g_master) std::printf(__VA_ARGS__); } while (0)
#define OUTS(text) do { if (g_master) std::puts(text); } while (0)

[[noreturn]] void fatal(const std::string& msg) {   // FatalError << ... << abort(FatalError)
    if (g_comm.size > 1) std::fprintf(stdout, "\n\n[%d] --> FOAM FATAL ERROR: \n%s\n\nFOAM aborting\n\n", g_comm.rank, msg.c_str());
    else std::fprintf(stdout, "\n\n--> FOAM FATAL ERROR: \n%s\n\nFOAM aborting\n\n", msg.c_str());
    std::fflush(stdout);
    std::_Exit(1);          // the parent of a -parallel run stops the other ranks
}

struct Options {
    std::map<std::string, std::string> kv;
    bool parallel = false;
    std::string caseDir = ".";
    bool found(const std::string& k) const { return kv.count(k) != 0; }
    double getD(const std::string& k, double def) const {
        if (!found(k)) return def;
        char* e = nullptr;
        const double v = std::strtod(kv.at(k).c_str(), &e);
        if (e == kv.at(k).c_str()) fatal("Bad value for option -" + k + ": " + kv.at(k));
        return v;
    }
    long getL(const std::string& k, long def) const {
        if (!found(k)) return def;
        char* e = nullptr;
        const long v = std::strtol(kv.at(k).c_str(), &e, 10);
        if (e == kv.at(k).c_str()) fatal("Bad value for option -" + k + ": " + kv.at(k));
        return v;
    }
    bool getB(const std::string& k, bool def) const {   // OpenFOAM Switch words
        if (!found(k)) return def;
        std::string v = kv.at(k);
        std::transform(v.begin(), v.end(), v.begin(), ::tolower);
        if (v == "true" || v == "on" || v == "yes" || v == "y" || v == "t" || v == "1") return true;
        if (v == "false" || v == "off" || v == "no" || v == "n" || v == "f" || v == "0" || v == "none") return false;
        fatal("Bad bool value for option -" + k + ": " + kv.at(k));
    }
};

// the options main() registers with argList::addOption (SM.C:1642-1784)
const char* kValueOptions[] = {"time", "centroidalIters", "maxStepLength", "relStepFrac", "edgeAngleConstraint",
                               "faceAngleConstraint", "minEdgeLength", "totalMinFreeze", "minAngle", "maxAngle",
                               "layerMaxBlendingFraction", "layerEdgeLength", "layerExpansionRatio", "minLayers",
                               "maxLayers", "layerPatches", "smoothingPatches", "internalSmoothingBlendingFraction",
                               "relTol", "writeInterval", "case", "writeFormat", "device"};

Options parseArgs(int argc, char** argv) {
    Options o;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-parallel") { o.parallel = true; continue; }
        if (a == "-help" || a == "-h") {
            std::puts("Usage: smoothMesh [-case dir] [-parallel] [-time t] [-centroidalIters n] [-relTol x] [-minEdgeLength x]\n"
                      "       [-maxStepLength x] [-relStepFrac x] [-totalMinFreeze b] [-edgeAngleConstraint b] [-faceAngleConstraint b]\n"
                      "       [-minAngle deg] [-maxAngle deg] [-writeInterval n] [-writeFormat ascii|binary] [-device n]\n"
                      "       [-layerPatches '(p1 \"re.*\")' -layerMaxBlendingFraction x -layerEdgeLength x -layerExpansionRatio x\n"
                      "        -minLayers n -maxLayers n]\n"
                      "       [-smoothingPatches '(p1 \"re.*\")' -internalSmoothingBlendingFraction x]   (boundary point smoothing, with\n"
                      "        constant/geometry/targetSurfaces.obj + initEdges.obj [+ targetEdges.obj])\n"
                      "Move internal mesh points to increase mesh quality (MI355X engine)");
            std::exit(0);
        }
        if (a.size() < 2 || a[0] != '-') fatal("Wrong argument " + a);
        const std::string key = a.substr(1);
        bool known = false;
        for (const char* k : kValueOptions) known |= (key == k);
        if (!known) fatal("Wrong option " + a);
        if (i + 1 >= argc) fatal("Option " + a + " requires an argument");
        o.kv[key] = argv[++i];
    }
    if (o.found("case")) o.caseDir = o.kv["case"];
    return o;
}

// minimal system/controlDict access: "key value;" entries at top level
std::map<std::string, std::string> readControlDict(const std::string& file) {
    std::map<std::string, std::string> d;
    FILE* f = std::fopen(file.c_str(), "rb");
    if (!f) return d;
    std::string s;
    char buf[4096];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    std::fclose(f);
    // strip comments
    std::string t;
    for (size_t i = 0; i < s.size(); ++i) {
        if (s[i] == '/' && i + 1 < s.size() && s[i + 1] == '/') { while (i < s.size() && s[i] != '\n') ++i; t.push_back('\n'); continue; }
        if (s[i] == '/' && i + 1 < s.size() && s[i + 1] == '*') { i += 2; while (i + 1 < s.size() && !(s[i] == '*' && s[i + 1] == '/')) ++i; ++i; continue; }
        t.push_back(s[i]);
    }
    int depth = 0;
    size_t i = 0;
    while (i < t.size()) {
        if (t[i] == '{') { ++depth; ++i; continue; }
        if (t[i] == '}') { --depth; ++i; continue; }
        if (depth == 0 && (std::isalpha((unsigned char)t[i]))) {
            size_t j = i;
            while (j < t.size() && !std::isspace((unsigned char)t[j]) && t[j] != ';' && t[j] != '{') ++j;
            const std::string key = t.substr(i, j - i);
            size_t k = j;
            while (k < t.size() && std::isspace((unsigned char)t[k])) ++k;
            if (k < t.size() && t[k] == '{') { i = k; continue; }
            size_t e = t.find(';', k);
            if (e == std::string::npos) break;
            std::string v = t.substr(k, e - k);
            while (!v.empty() && std::isspace((unsigned char)v.back())) v.pop_back();
            d[key] = v;
            i = e + 1;
            continue;
        }
        ++i;
    }
    return d;
}

std::vector<std::pair<double, std::string>> listTimes(const std::string& dir) {
    std::vector<std::pair<double, std::string>> t;
    DIR* d = ::opendir(dir.c_str());
    if (!d) return t;
    while (dirent* e = ::readdir(d)) {
        const std::string n = e->d_name;
        if (n == "." || n == "..") continue;
        char* end = nullptr;
        const double v = std::strtod(n.c_str(), &end);
        if (end != n.c_str() && *end == '\0' && dirExists(dir + "/" + n)) t.push_back({v, n});
    }
    ::closedir(d);
    std::sort(t.begin(), t.end());
    return t;
}

std::string timeName(double t) {   // timeFormat general; timePrecision 6 (testcase/system/controlDict:34-36)
    char b[64];
    std::snprintf(b, sizeof b, "%.6g", t);
    return b;
}

// SM.C:40-91 findInternalMeshPoints
std::vector<uint8_t> findInternalMeshPoints(const PolyMeshData& m) {
    std::vector<uint8_t> in((size_t)m.nPoints(), 1);
    for (const auto& p : m.patches) {
        if (p.type == "processor") continue;
        if (p.type == "empty") fatal("Smoothing of non-3D meshes (meshes with type empty patches) is not supported");
        for (int32_t f = p.startFace; f < p.startFace + p.nFaces; ++f)
            for (int32_t k = m.faceOffsets[f]; k < m.faceOffsets[f + 1]; ++k) in[(size_t)m.facePoints[k]] = 0;
    }
    return in;
}

void check(int rc, const char* what) {
    if (rc) fatal(std::string(what) + ": " + smgpu_last_error());
}
#define HIPCHK(e) do { hipError_t r__ = (e); if (r__ != hipSuccess) fatal(std::string(#e) + ": " + hipGetErrorString(r__)); } while (0)
#define NCCLCHK(e) do { ncclResult_t r__ = (e); if (r__ != ncclSuccess) fatal(std::string(#e) + ": " + ncclGetErrorString(r__)); } while (0)

struct Rank {
    std::string root;            // case dir (serial) or processorN dir
    PolyMeshData mesh;
    std::vector<uint8_t> internal;
    std::vector<int64_t> pointProc;            // local point -> global id (pointProcAddressing) or the label labelsFromPatches gives it
    smgpu_handle* h = nullptr;
    int device = 0;
    // halo
    std::vector<int32_t> sharedLocal, sendShared, combOff, combSlots;
    std::vector<int64_t> sharedGlobal;
    std::vector<std::vector<int>> sharersOf;            // per shared point: the ranks of its group, ascending, this rank included
    std::vector<int> peerCount, peerSendBase;
    double *sendA = nullptr, *recvA = nullptr, *localStats = nullptr, *sendL = nullptr, *recvL = nullptr;
    int32_t *sendF = nullptr, *recvF = nullptr;
    int nSend = 0;
};

std::string findInstance(const std::string& root, const std::vector<std::pair<double, std::string>>& times, double startValue,
                         bool startIsConstant, const std::string& file) {
    if (!startIsConstant)
        for (auto it = times.rbegin(); it != times.rend(); ++it)
            if (it->first <= startValue + 1e-12 && (fileExists(root + "/" + it->second + "/polyMesh/" + file) || fileExists(root + "/" + it->second + "/polyMesh/" + file + ".gz")))
                return root + "/" + it->second + "/polyMesh";
    return root + "/constant/polyMesh";
}

// Who shares which point with this rank, as OpenFOAM's globalPoints finds it: the copies of a point on the two sides of a
// PROCESSOR PATCH are the same point, and so is everything connected through such pairs (a rank has one local point per mesh
// point, which joins all its patches) -- and nothing else: the two sides of a baffle (createBaffles, the reference's testcase6)
// on different ranks are different shared points, or none (smoothmesh_amd/decompose.py:shared_point_components is the same rule).
// lists[r] = rank r's processor patches, flattened: {neighbour, count, ids ascending ...}*
// -> for every shared point of rank r (global id): the ranks of its group, ascending, r included
std::map<int64_t, std::vector<int>> sharedGroupsOf(int r, int n, const std::vector<std::vector<int64_t>>& lists) {
    std::vector<std::map<int, std::vector<int64_t>>> patch((size_t)n);
    std::vector<std::vector<int64_t>> ids((size_t)n);
    for (int o = 0; o < n; ++o) {
        const auto& v = lists[(size_t)o];
        for (size_t k = 0; k + 1 < v.size();) {
            const int nb = (int)v[k]; const size_t c = (size_t)v[k + 1];
            auto& dst = patch[(size_t)o][nb];
            dst.insert(dst.end(), v.begin() + (ptrdiff_t)(k + 2), v.begin() + (ptrdiff_t)(k + 2 + c));
            k += 2 + c;
        }
        for (auto& kv : patch[(size_t)o]) {
            std::sort(kv.second.begin(), kv.second.end());
            kv.second.erase(std::unique(kv.second.begin(), kv.second.end()), kv.second.end());
            ids[(size_t)o].insert(ids[(size_t)o].end(), kv.second.begin(), kv.second.end());
        }
        std::sort(ids[(size_t)o].begin(), ids[(size_t)o].end());
        ids[(size_t)o].erase(std::unique(ids[(size_t)o].begin(), ids[(size_t)o].end()), ids[(size_t)o].end());
    }
    std::vector<size_t> base((size_t)n + 1, 0);
    for (int o = 0; o < n; ++o) base[(size_t)o + 1] = base[(size_t)o] + ids[(size_t)o].size();
    std::vector<size_t> parent(base[(size_t)n]);
    for (size_t i = 0; i < parent.size(); ++i) parent[i] = i;
    auto find = [&](size_t a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
    auto node = [&](int o, int64_t g) { return base[(size_t)o] + (size_t)(std::lower_bound(ids[(size_t)o].begin(), ids[(size_t)o].end(), g) - ids[(size_t)o].begin()); };
    for (int a = 0; a < n; ++a)
        for (const auto& kv : patch[(size_t)a]) {
            const int b = kv.first;
            if (b <= a || b >= n) continue;
            const auto it = patch[(size_t)b].find(a);
            if (it == patch[(size_t)b].end()) continue;
            std::vector<int64_t> common;
            std::set_intersection(kv.second.begin(), kv.second.end(), it->second.begin(), it->second.end(), std::back_inserter(common));
            for (int64_t g : common) {
                const size_t x = find(node(a, g)), y = find(node(b, g));
                if (x != y) parent[x] = y;
            }
        }
    std::map<size_t, std::vector<int>> members;        // root -> ranks (ascending: filled in rank order), only roots of r's nodes
    std::map<size_t, int64_t> rootOfMine;
    for (size_t i = 0; i < ids[(size_t)r].size(); ++i) rootOfMine[find(base[(size_t)r] + i)] = ids[(size_t)r][i];
    for (int o = 0; o < n; ++o)
        for (size_t i = 0; i < ids[(size_t)o].size(); ++i) {
            const size_t root = find(base[(size_t)o] + i);
            if (rootOfMine.count(root)) members[root].push_back(o);
        }
    std::map<int64_t, std::vector<int>> out;
    for (const auto& kv : members)
        if (kv.second.size() >= 2) out[rootOfMine.at(kv.first)] = kv.second;
    return out;
}

// the shared-point tables of this rank from its points' groups (what smoothmesh_amd/halo.py:HaloTables builds)
void buildHalo(Rank& K, int r, int n, const std::map<int64_t, std::vector<int>>& groups) {
    std::vector<std::vector<int64_t>> shared(n);
    std::set<int64_t> all;
    K.sharersOf.clear();
    for (const auto& kv : groups) {                    // (ascending global id)
        all.insert(kv.first);
        K.sharersOf.push_back(kv.second);
        for (int o : kv.second) if (o != r) shared[(size_t)o].push_back(kv.first);
    }
    K.sharedGlobal.assign(all.begin(), all.end());
    std::map<int64_t, int32_t> g2l;
    for (int32_t p = 0; p < K.mesh.nPoints(); ++p) g2l[K.pointProc[(size_t)p]] = p;
    K.sharedLocal.clear();
    for (int64_t g : K.sharedGlobal) K.sharedLocal.push_back(g2l.at(g));
    K.peerCount.assign(n, 0);
    K.peerSendBase.assign(n, 0);
    K.sendShared.clear();
    std::vector<std::vector<std::pair<int, int>>> per(K.sharedGlobal.size());   // (rank, slot)
    for (size_t i = 0; i < per.size(); ++i) per[i].push_back({r, -1});
    int run = 0;
    for (int o = 0; o < n; ++o) {
        K.peerSendBase[o] = run;
        K.peerCount[o] = (int)shared[o].size();
        for (size_t k = 0; k < shared[o].size(); ++k) {
            const int32_t idx = (int32_t)(std::lower_bound(K.sharedGlobal.begin(), K.sharedGlobal.end(), shared[o][k]) - K.sharedGlobal.begin());
            K.sendShared.push_back(idx);
            per[(size_t)idx].push_back({o, run + (int)k});
        }
        run += (int)shared[o].size();
    }
    K.nSend = run;
    K.combOff.assign(1, 0);
    K.combSlots.clear();
    for (auto& v : per) {
        std::sort(v.begin(), v.end());
        for (auto& pr : v) K.combSlots.push_back(pr.second);
        K.combOff.push_back((int32_t)K.combSlots.size());
    }
}

// Sub-domains WITHOUT pointProcAddressing (a mesh made in parallel, e.g. by snappyHexMesh -parallel: the files decomposePar writes
// are absent or stale): the copies of a point are matched through the processor patches themselves, as OpenFOAM's globalPoints
// does -- face i of rank a's patch to b is face i of b's patch to a, reversed about its first vertex (face::reverseFace:
// the normal points out of either domain), so vertex k of one is vertex (n - k) % n of the other.  Every (rank, local point) on
// a processor patch gets the label of its connected component, (lowest rank << 40) | that rank's local id; the other points
// keep (own rank << 40) | local id.  The labels stand in for the global point ids everywhere below.
// own = this rank's patches flattened: {neighbour, nFaces, {n, local ids ...} per face}*
std::vector<int64_t> processorPatchFaces(const Rank& K) {
    std::vector<int64_t> out;
    const auto& m = K.mesh;
    for (const auto& p : m.patches)
        if (p.type == "processor") {
            out.push_back((int64_t)p.neighbProcNo);
            out.push_back((int64_t)p.nFaces);
            for (int32_t f = p.startFace; f < p.startFace + p.nFaces; ++f) {
                out.push_back((int64_t)(m.faceOffsets[f + 1] - m.faceOffsets[f]));
                for (int32_t k = m.faceOffsets[f]; k < m.faceOffsets[f + 1]; ++k) out.push_back((int64_t)m.facePoints[k]);
            }
        }
    return out;
}
std::string labelsFromPatches(Rank& K, int r, int n, const std::vector<std::vector<int64_t>>& all) {
    struct PatchFaces { std::vector<size_t> at; };                  // per face: where its {n, ids...} record starts in all[rank]
    std::vector<std::map<int, PatchFaces>> patch((size_t)n);
    for (int o = 0; o < n; ++o) {
        const auto& v = all[(size_t)o];
        for (size_t k = 0; k + 1 < v.size();) {
            const int nb = (int)v[k]; const int64_t nF = v[k + 1];
            k += 2;
            PatchFaces& pf = patch[(size_t)o][nb];
            for (int64_t f = 0; f < nF; ++f) { pf.at.push_back(k); k += 1 + (size_t)v[k]; }
        }
    }
    std::map<std::pair<int, int64_t>, std::pair<int, int64_t>> parent;   // (rank, local point) -> parent; the root is the lowest (rank, id)
    typedef std::pair<int, int64_t> Node;
    std::function<Node(Node)> find = [&](Node a) {
        auto it = parent.find(a);
        if (it == parent.end()) { parent[a] = a; return a; }
        if (it->second == a) return a;
        const Node root = find(it->second);
        parent[a] = root;
        return root;
    };
    for (int a = 0; a < n; ++a)
        for (const auto& kv : patch[(size_t)a]) {
            const int b = kv.first;
            if (b <= a) continue;
            if (b >= n) return "processor patch to a rank that does not exist";
            const auto it = patch[(size_t)b].find(a);
            if (it == patch[(size_t)b].end() || it->second.at.size() != kv.second.at.size())
                return "processor patches " + std::to_string(a) + " <-> " + std::to_string(b) + " do not match";
            for (size_t f = 0; f < kv.second.at.size(); ++f) {
                const int64_t* fa = &all[(size_t)a][kv.second.at[f]];
                const int64_t* fb = &all[(size_t)b][it->second.at[f]];
                const int64_t nv = fa[0];
                if (fb[0] != nv) return "processor patches " + std::to_string(a) + " <-> " + std::to_string(b) + ": face sizes differ";
                for (int64_t k = 0; k < nv; ++k) {
                    const Node x = find(Node(a, fa[1 + k])), y = find(Node(b, fb[1 + (nv - k) % nv]));
                    if (x != y) { if (x < y) parent[y] = x; else parent[x] = y; }
                }
            }
        }
    K.pointProc.resize((size_t)K.mesh.nPoints());
    std::map<Node, int32_t> firstOfRoot;      // a rank has ONE local point per mesh point: two of them in one component mean the
                                              // patches do not pair up the way globalPoints assumes (rotated faces, stale processorN data)
    for (int32_t p = 0; p < K.mesh.nPoints(); ++p) {
        Node root(r, (int64_t)p);
        if (parent.count(root)) {
            root = find(root);
            const auto ins = firstOfRoot.emplace(root, p);
            if (!ins.second)
                return "processor patches of rank " + std::to_string(r) + ": local points " + std::to_string(ins.first->second) + " and " + std::to_string(p) +
                       " are matched to the same shared point (faces of a processor patch pair are not in corresponding order)";
        }
        K.pointProc[(size_t)p] = ((int64_t)root.first << 40) | root.second;
    }
    return "";
}

// The coordinates behind processorPatchFaces, in the same order (three doubles per listed vertex) ...
std::vector<double> processorPatchCoords(const Rank& K) {
    std::vector<double> out;
    const auto& m = K.mesh;
    for (const auto& p : m.patches)
        if (p.type == "processor")
            for (int32_t f = p.startFace; f < p.startFace + p.nFaces; ++f)
                for (int32_t k = m.faceOffsets[f]; k < m.faceOffsets[f + 1]; ++k)
                    for (int c = 0; c < 3; ++c) out.push_back(m.points[3 * (size_t)m.facePoints[k] + (size_t)c]);
    return out;
}
// ... and the check nothing else makes: face i of rank a's patch to b is face i of b's patch to a reversed about its first vertex
// (OpenFOAM's processorPolyPatch convention, what globalPoints and labelsFromPatches rely on), so the matched vertices must
// coincide -- within OpenFOAM's own matchTolerance (1e-4 of the face's extent).  A patch pair with rotated faces or processorN
// directories from two different decompositions would otherwise give wrong halo tables without a word.
std::string checkProcessorPatchCoords(int r, int n, const std::vector<std::vector<int64_t>>& faces, const std::vector<std::vector<double>>& coords) {
    struct Rec { size_t at, co; };                                    // per face: start of its {n, ids...} record / of its coordinates
    std::vector<std::map<int, std::vector<Rec>>> patch((size_t)n);
    for (int o = 0; o < n; ++o) {
        const auto& v = faces[(size_t)o];
        size_t co = 0;
        for (size_t k = 0; k + 1 < v.size();) {
            const int nb = (int)v[k]; const int64_t nF = v[k + 1];
            k += 2;
            auto& pf = patch[(size_t)o][nb];
            for (int64_t f = 0; f < nF; ++f) { pf.push_back(Rec{k, co}); co += 3 * (size_t)v[k]; k += 1 + (size_t)v[k]; }
        }
        if (co != coords[(size_t)o].size()) return "processor patch coordinates of rank " + std::to_string(o) + " do not match its face list";
    }
    for (const auto& kv : patch[(size_t)r]) {
        const int b = kv.first;
        if (b < 0 || b >= n) return "processor patch to a rank that does not exist";
        const auto it = patch[(size_t)b].find(r);
        if (it == patch[(size_t)b].end() || it->second.size() != kv.second.size())
            return "processor patches " + std::to_string(r) + " <-> " + std::to_string(b) + " do not match";
        for (size_t f = 0; f < kv.second.size(); ++f) {
            const int64_t nv = faces[(size_t)r][kv.second[f].at];
            if (faces[(size_t)b][it->second[f].at] != nv) return "processor patches " + std::to_string(r) + " <-> " + std::to_string(b) + ": face sizes differ";
            const double* xa = &coords[(size_t)r][kv.second[f].co];
            const double* xb = &coords[(size_t)b][it->second[f].co];
            double ext = 0.0;
            for (int64_t k = 1; k < nv; ++k) {
                double d2 = 0.0;
                for (int c = 0; c < 3; ++c) { const double d = xa[3 * k + c] - xa[c]; d2 += d * d; }
                ext = std::max(ext, std::sqrt(d2));
            }
            for (int64_t k = 0; k < nv; ++k) {
                const int64_t kb = (nv - k) % nv;
                double d2 = 0.0;
                for (int c = 0; c < 3; ++c) { const double d = xa[3 * k + c] - xb[3 * kb + c]; d2 += d * d; }
                if (!(std::sqrt(d2) <= 1e-4 * ext))
                    return "processor patches " + std::to_string(r) + " <-> " + std::to_string(b) + ": face " + std::to_string(f) + " vertex " + std::to_string(k) +
                           " lies " + std::to_string(std::sqrt(d2)) + " away from its copy on the other side (face extent " + std::to_string(ext) +
                           "): the two sides are not in corresponding order, or the processor directories are not from one decomposition";
            }
        }
    }
    return "";
}

// this rank's processor patches as {neighbour, count, global point ids ascending ...}* (sharedGroupsOf)
std::vector<int64_t> processorPatchLists(const Rank& K) {
    std::vector<int64_t> out;
    const auto& m = K.mesh;
    for (const auto& p : m.patches)
        if (p.type == "processor") {
            std::set<int64_t> s;
            for (int32_t f = p.startFace; f < p.startFace + p.nFaces; ++f)
                for (int32_t k = m.faceOffsets[f]; k < m.faceOffsets[f + 1]; ++k) s.insert(K.pointProc[(size_t)m.facePoints[k]]);
            out.push_back((int64_t)p.neighbProcNo);
            out.push_back((int64_t)s.size());
            out.insert(out.end(), s.begin(), s.end());
        }
    return out;
}

}  // namespace

int main(int argc, char** argv) {
    const auto t0 = std::chrono::steady_clock::now();
    // where the ClockTime of the last line goes (the reference prints the total only, SM.C:2439): read the case, set the engine up
    // (addressing, tile tables, upload, halo tables), the smoothing loop itself, mesh output
    double tRead = 0.0, tCreate = 0.0, tLoop = 0.0, tWrite = 0.0;
    auto secondsSince = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
    const Options opt = parseArgs(argc, argv);
    const std::string& cd = opt.caseDir;
    // SMOOTHMESH_TIMELINE=1: where this process's wall time goes, one stderr line per stage (seconds since main started)
    const bool timeline = std::getenv("SMOOTHMESH_TIMELINE") && std::atoi(std::getenv("SMOOTHMESH_TIMELINE")) > 0;
    auto mark = [&](const char* what) { if (timeline) std::fprintf(stderr, "[smoothMesh %8.3f s] %s\n", secondsSince(t0), what); };

    // createTime.H: deltaT sanity (SM.C:1805-1812)
    const auto control = readControlDict(cd + "/system/controlDict");
    const double deltaT = control.count("deltaT") ? std::atof(control.at("deltaT").c_str()) : 1.0;
    if (deltaT < VSMALL) fatal("Time step (deltaT) value " + std::to_string(deltaT) + " specified in controlDict is too small");
    const bool binary = opt.found("writeFormat") ? (opt.kv.at("writeFormat") == "binary")
                                                 : (control.count("writeFormat") && control.at("writeFormat") == "binary");
    int writePrecision = control.count("writePrecision") ? std::atoi(control.at("writePrecision").c_str()) : 6;
    writePrecision = std::max(10, writePrecision);   // SM.C:2425
    if (control.count("writeCompression")) {          // on / true / yes / compressed
        const std::string wc = control.at("writeCompression");
        setWriteCompression(wc == "on" || wc == "true" || wc == "yes" || wc == "compressed");
    }

    // sub-domain roots.  -parallel: one process per processorN/ directory, started here, before anything touches HIP
    std::vector<Rank> R(1);
    int nRanks = 1;
    if (opt.parallel) {
        nRanks = 0;
        while (dirExists(cd + "/processor" + std::to_string(nRanks))) ++nRanks;
        if (nRanks == 0) fatal("-parallel: no processor0 directory in " + cd + " (run decomposePar first)");
        g_comm.launch(nRanks);                 // returns in the ranks only
        g_master = g_comm.master();
        R[0].root = cd + "/processor" + std::to_string(g_comm.rank);
    } else {
        R[0].root = cd;
    }
    const int myRank = g_comm.rank;
    Rank& K0 = R[0];
    // The HIP runtime comes up (device discovery, context, first allocation: a few tenths of a second per process) on a thread of its
    // own while this one reads the case -- in every rank, after the ranks were started (nothing touches HIP before that)
    std::atomic<int> warmDevices{-1};
    std::thread hipWarm([&warmDevices, &opt, myRank] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { warmDevices = 0; return; }
        if (hipSetDevice(((int)opt.getL("device", 0) + myRank) % n) == hipSuccess) {
            // (the runtime builds its queues and loads its copy / fill kernels at their first use, ~0.2 s: use each once)
            void* p = nullptr;
            char b[64] = {0};
            if (hipMalloc(&p, 1 << 20) == hipSuccess) {
                (void)hipMemset(p, 0, 1 << 20);
                (void)hipMemcpy(p, b, sizeof b, hipMemcpyHostToDevice);
                (void)hipMemcpy(b, p, sizeof b, hipMemcpyDeviceToHost);
                (void)hipDeviceSynchronize();
                (void)hipFree(p);
            }
            (void)hipGetLastError();
        }
        warmDevices = n;
    });
    struct JoinWarm { std::thread& t; ~JoinWarm() { if (t.joinable()) t.join(); } } joinWarm{hipWarm};

    // start time (SM.C:1791-1803; controlDict startFrom latestTime)
    const auto times = listTimes(R[0].root);
    bool startIsConstant = false;
    double startValue = 0.0;
    if (opt.found("time")) {
        if (opt.kv.at("time") == "constant") startIsConstant = true;
        else startValue = opt.getD("time", 0.0);
    } else if (!times.empty()) startValue = times.back().first;
    else startIsConstant = true;

    OUT("smoothMesh (MI355X engine %s)\nCase: %s%s\n", smgpu_version(), cd.c_str(), opt.parallel ? "  [parallel]" : "");
    if (opt.parallel) OUT("nProcs : %d   (one process per sub-domain, pid of the master %d)\n", nRanks, (int)getpid());
    OUT("Create mesh for time = %s\n\n", startIsConstant ? "constant" : timeName(startValue).c_str());

    try {
        for (Rank& K : R) {
            const std::string meshDir = findInstance(K.root, listTimes(K.root), startValue, startIsConstant, "faces");
            const std::string ptsDir = findInstance(K.root, listTimes(K.root), startValue, startIsConstant, "points");
            { const auto tr = std::chrono::steady_clock::now(); readPolyMesh(meshDir, ptsDir == meshDir ? "" : ptsDir, K.mesh); tRead += secondsSince(tr); }
            mark("case read");
            K.internal = findInternalMeshPoints(K.mesh);
            mark("internal points found");
            if (opt.parallel) {
                // shared points are matched by decomposePar's global ids where the file is there, and through the processor patches
                // themselves where it is not (labelsFromPatches; SMGPU_MATCH_BY_PATCHES=1 forces that)
                const std::string ppa = K.root + "/constant/polyMesh/pointProcAddressing";
                const char* force = std::getenv("SMGPU_MATCH_BY_PATCHES");
                if (!(force && std::atoi(force)) && (fileExists(ppa) || fileExists(ppa + ".gz"))) {
                    std::vector<int32_t> ids;
                    readLabelList(ppa, ids);
                    if ((int32_t)ids.size() != K.mesh.nPoints()) fatal(ppa + ": size does not match the number of points");
                    K.pointProc.assign(ids.begin(), ids.end());
                }
            }
        }
    } catch (const std::exception& e) { fatal(e.what()); }

    // patches for the boundary layer treatment, getPatchIdsForOption SM.C:1442-1471 / 1823-1833: a wordRe list,
    // "(name1 name2 \"regex.*\")" or a single word; quoted entries are regular expressions.  Every sub-domain carries
    // the original patches (possibly empty) plus its processor patches, so the names are matched per sub-domain.
    typedef std::vector<std::pair<std::string, bool>> WordRes;   // (word, is a regular expression)
    auto parseWordRes = [&](const std::string& optName, std::string v) {
        WordRes words;
        for (char& ch : v) if (ch == '(' || ch == ')') ch = ' ';
        size_t i = 0;
        while (i < v.size()) {
            while (i < v.size() && std::isspace((unsigned char)v[i])) ++i;
            if (i >= v.size()) break;
            bool isRe = false;
            std::string tok;
            if (v[i] == '"') { isRe = true; ++i; while (i < v.size() && v[i] != '"') tok.push_back(v[i++]); ++i; }
            else while (i < v.size() && !std::isspace((unsigned char)v[i])) tok.push_back(v[i++]);
            if (isRe) {
                try { (void)std::regex(tok, std::regex::extended); }
                catch (const std::regex_error&) { fatal("-" + optName + ": bad regular expression \"" + tok + "\""); }
            }
            words.push_back({tok, isRe});
        }
        return words;
    };
    auto matchesAny = [](const WordRes& words, const std::string& name) {
        for (const auto& w : words)
            if (w.second ? std::regex_match(name, std::regex(w.first, std::regex::extended)) : (name == w.first)) return true;
        return false;
    };
    WordRes layerWords;
    if (opt.found("layerPatches")) layerWords = parseWordRes("layerPatches", opt.kv.at("layerPatches"));
    bool anyLayerPatch = false;
    std::vector<std::vector<uint8_t>> isLayerPatchOf(R.size());
    for (size_t r = 0; r < R.size(); ++r) {
        const auto& patches = R[r].mesh.patches;
        isLayerPatchOf[r].assign(patches.size(), 0);
        for (size_t p = 0; p < patches.size(); ++p)
            if (matchesAny(layerWords, patches[p].name)) { isLayerPatchOf[r][p] = 1; anyLayerPatch = true; }
    }
    anyLayerPatch = g_comm.reduceOr(anyLayerPatch);
    if (anyLayerPatch) OUT("Patches for boundary layer treatment: %s\n", opt.kv.at("layerPatches").c_str());
    else OUTS("Patches for boundary layer treatment: none");
    const double layerMaxBlendingFraction = opt.getD("layerMaxBlendingFraction", 0.3);
    const bool doLayerTreatment = anyLayerPatch && layerMaxBlendingFraction > SMALL;   // SM.C:2024-2028
    // smoothing patches: every patch unless -smoothingPatches says otherwise (SM.C:1835-1853)
    const std::string smoothingOpt = opt.found("smoothingPatches") ? opt.kv.at("smoothingPatches") : std::string("(\".*\")");
    const WordRes smoothingWords = parseWordRes("smoothingPatches", smoothingOpt);
    bool anySmoothingPatch = false;
    std::vector<std::vector<uint8_t>> isSmoothingPatchOf(R.size());
    for (size_t r = 0; r < R.size(); ++r) {
        const auto& patches = R[r].mesh.patches;
        isSmoothingPatchOf[r].assign(patches.size(), 0);
        for (size_t p = 0; p < patches.size(); ++p)
            if (matchesAny(smoothingWords, patches[p].name)) { isSmoothingPatchOf[r][p] = 1; anySmoothingPatch = true; }
    }
    anySmoothingPatch = g_comm.reduceOr(anySmoothingPatch);
    if (anySmoothingPatch) OUT("Patches for boundary point smoothing: %s\n", smoothingOpt.c_str());
    else OUTS("Patches for boundary point smoothing: none");
    const double internalSmoothingBlendingFraction = opt.getD("internalSmoothingBlendingFraction", 0.0);   // SM.C:1907

    if (doLayerTreatment) OUTS("Enabled boundary layer treatment\n");
    else OUTS("Boundary layer treatment is disabled. Either no layerPatches were specified or boundaryMaxBlendingFraction is zero\n");

    // classification lists of a previous run (labelIOLists <time>/isCornerPoint, <time>/isFeatureEdgePoint, SM.C:2039-2077)
    const std::string startName = startIsConstant ? std::string("constant") : timeName(startValue);
    std::vector<std::vector<int32_t>> isCornerPointIO(R.size()), isFeatureEdgePointIO(R.size());   // per sub-domain
    bool labelIOListsHaveData = false;
    for (size_t r = 0; r < R.size(); ++r) {
        auto readIfPresent = [&](const std::string& name, std::vector<int32_t>& out) {
            const std::string f = R[r].root + "/" + startName + "/" + name;
            if (!fileExists(f) && !fileExists(f + ".gz")) return;
            try { readLabelList(f, out); } catch (const std::exception& e) { fatal(e.what()); }
            if ((int32_t)out.size() != R[r].mesh.nPoints()) fatal(f + ": size does not match the number of points");
            for (int32_t v : out) labelIOListsHaveData = labelIOListsHaveData || v == 1;
        };
        readIfPresent("isCornerPoint", isCornerPointIO[r]);
        readIfPresent("isFeatureEdgePoint", isFeatureEdgePointIO[r]);
    }
    labelIOListsHaveData = g_comm.reduceOr(labelIOListsHaveData);
    if (labelIOListsHaveData) OUTS("Found corners and feature edges in isCornerPoint and isFeatureEdgePoint files\n");
    else OUTS("Did not find corners and feature edges in isCornerPoint and isFeatureEdgePoint files\n");

    // prerequisites of the boundary point smoothing, SM.C:2080-2093
    const std::string targetSurfacesFile = "constant/geometry/targetSurfaces.obj", initEdgesFile = "constant/geometry/initEdges.obj",
                      targetEdgesFile = "constant/geometry/targetEdges.obj";
    const bool doBoundarySmoothing = fileExists(cd + "/" + targetSurfacesFile) && (fileExists(cd + "/" + initEdgesFile) || labelIOListsHaveData) &&
                                     anySmoothingPatch;
    if (doBoundarySmoothing) OUTS("Enabled boundary point smoothing\n");
    else OUT("Boundary point smoothing is disabled. Missing smoothingPatches, or one or both of files:\n%s\n%s\n\n", targetSurfacesFile.c_str(), initEdgesFile.c_str());
    if (doLayerTreatment && !doBoundarySmoothing)   // SM.C:2095-2098
        OUTS("WARNING: Boundary layer treatment will be done without boundary point smoothing. This can result in distorted boundary cells.\n");

    // engine: this rank's sub-domain on its device
    mark("options and boundary set-up read");
    if (hipWarm.joinable()) hipWarm.join();
    mark("HIP runtime up");
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0) fatal("no HIP device available (this build has no CPU fallback)");
    const int dev0 = (int)opt.getL("device", 0);
    {
        Rank& K = K0;
        smgpu_mesh_desc d{};
        d.nPoints = K.mesh.nPoints(); d.nCells = K.mesh.nCells; d.nFaces = K.mesh.nFaces(); d.nInternalFaces = K.mesh.nInternalFaces();
        d.points = K.mesh.points.data(); d.faceOffsets = K.mesh.faceOffsets.data(); d.facePoints = K.mesh.facePoints.data();
        d.owner = K.mesh.owner.data(); d.neighbour = K.mesh.neighbour.data();
        d.isInternalPoint = K.internal.data(); d.isSmoothingSurfacePoint = nullptr;
        K.device = (dev0 + myRank) % nDev;
        d.device = K.device; d.stream = nullptr; d.useCallerStream = 0;
        { const auto tc = std::chrono::steady_clock::now(); check(smgpu_create(&d, &K.h), "smgpu_create"); tCreate += secondsSince(tc); }
        mark("engine created");
        // several ranks on one device (a debugging arrangement): every rank's persistent walk replay needs all of its workgroups
        // resident at once, so each takes its share of the chip (the engine's default is sized for a device of its own)
        if (nRanks > nDev) check(smgpu_set_device_share(K.h, (nRanks + nDev - 1) / nDev), "smgpu_set_device_share");
    }
    // transport of the per-iteration records between the ranks
    enum { TRANSPORT_RCCL, TRANSPORT_SHM, TRANSPORT_PUSH } transport = TRANSPORT_RCCL;
    ncclComm_t nccl = nullptr;
    hipStream_t engineStream = nullptr;
    struct PushHandles { char h[4][64]; } pushHandles{};   // recvA, recvL, recvF, flags
    void* pushFlags = nullptr;
    std::vector<void*> pushMapped, pushA, pushL, pushF, pushFl;
    std::vector<int32_t> pushCount, pushBase, pushIndex;
    if (opt.parallel) {
        const char* tv = std::getenv("SMOOTHMESH_TRANSPORT");
        if (tv && std::string(tv) == "shm") transport = TRANSPORT_SHM;
        else if (tv && std::string(tv) == "rccl") transport = TRANSPORT_RCCL;
        else if (tv && std::string(tv) == "push") transport = TRANSPORT_PUSH;   // peer stores (include/smgpu.h, smgpu_push_desc)
        else if (nDev < nRanks) {
            transport = TRANSPORT_SHM;
            OUT("WARNING: %d ranks on %d GPU(s): RCCL needs one device per rank; staging the shared-point records through host memory (debug transport)\n\n", nRanks, nDev);
        }
        if (transport == TRANSPORT_PUSH) OUTS("Shared-point records: peer stores into the ranks' mapped receive buffers (SMOOTHMESH_TRANSPORT=push)\n");
        void* vs = nullptr;
        check(smgpu_get_stream(K0.h, &vs), "smgpu_get_stream");
        engineStream = (hipStream_t)vs;
        HIPCHK(hipSetDevice(K0.device));
        if (transport == TRANSPORT_RCCL) {
            ncclUniqueId id;
            std::memset(&id, 0, sizeof id);
            if (g_master) NCCLCHK(ncclGetUniqueId(&id));
            id = g_comm.broadcast(id, 0);
            NCCLCHK(ncclCommInitRank(&nccl, nRanks, id, myRank));
            int seen = -1;
            NCCLCHK(ncclCommCount(nccl, &seen));           // what the communicator itself says about its size
            if (seen != nRanks) fatal("RCCL communicator has " + std::to_string(seen) + " ranks, the case has " + std::to_string(nRanks));
            OUT("Shared-point records: RCCL send / recv groups on the engines' streams, communicator of %d ranks\n", seen);
        }
    }

    // getMeshStats + defaults (SM.C:1857-1918)
    double meshMinEdgeLength = 1e300, meshMaxEdgeLength = 0.0;
    for (Rank& K : R) {
        double a, b;
        check(smgpu_mesh_stats(K.h, &a, &b), "smgpu_mesh_stats");
        meshMinEdgeLength = std::min(meshMinEdgeLength, a);
        meshMaxEdgeLength = std::max(meshMaxEdgeLength, b);
    }
    meshMinEdgeLength = g_comm.reduceMin(meshMinEdgeLength);   // returnReduce, SM.C:1527-1535
    meshMaxEdgeLength = g_comm.reduceMax(meshMaxEdgeLength);
    smgpu_params prm{};
    prm.minEdgeLength = opt.getD("minEdgeLength", 0.5 * meshMinEdgeLength);
    prm.maxStepLength = opt.getD("maxStepLength", 0.3 * prm.minEdgeLength);
    if (prm.maxStepLength > 0.5 * prm.minEdgeLength)
        OUTS("WARNING: The maximum allowed step length is more than half of the minimum edge length! This may cause unstability in smoothing.\n");
    prm.relStepFrac = opt.getD("relStepFrac", 0.5);
    prm.totalMinFreeze = opt.getB("totalMinFreeze", false);
    prm.minAngle = opt.getD("minAngle", 35.0);
    prm.maxAngle = opt.getD("maxAngle", 160.0);
    prm.edgeAngleConstraint = opt.getB("edgeAngleConstraint", true);
    prm.faceAngleConstraint = opt.getB("faceAngleConstraint", true);
    const double relTol = opt.getD("relTol", 0.02);
    const long centroidalIters = opt.getL("centroidalIters", 1000);
    const long writeInterval = opt.getL("writeInterval", centroidalIters);
    if (writeInterval <= 0) fatal("writeInterval must be positive");

    // parameter echo, SM.C:1933-1975
    OUTS("Applying following parameter values in smoothing:");
    OUT("    centroidalIters        %ld\n    relTol                 %g\n    minEdgeLength          %g\n", centroidalIters, relTol, prm.minEdgeLength);
    OUT("    maxStepLength          %g\n    relStepFrac            %g\n    totalMinFreeze         %d\n", prm.maxStepLength, prm.relStepFrac, prm.totalMinFreeze);
    if (prm.edgeAngleConstraint) OUT("    edgeAngleConstraint    true\n    minAngle               %g\n", prm.minAngle);
    else OUTS("    edgeAngleConstraint    false (edge min angle quality constraint is NOT applied)");
    if (prm.faceAngleConstraint) OUT("    faceAngleConstraint    true\n    minAngle               %g\n    maxAngle               %g\n", prm.minAngle, prm.maxAngle);
    else OUTS("    faceAngleConstraint    false (face angle quality constraints are NOT applied)");
    const double layerEdgeLength = opt.getD("layerEdgeLength", prm.minEdgeLength);       // SM.C:1895-1905
    const double layerExpansionRatio = opt.getD("layerExpansionRatio", 1.3);
    const long minLayers = opt.getL("minLayers", 1), maxLayers = opt.getL("maxLayers", 4);
    if (layerMaxBlendingFraction > SMALL)
        OUT("    layerMaxBlendingFraction %g\n    layerEdgeLength          %g\n    layerExpansionRatio      %g\n    minLayers                %ld\n"
                    "    maxLayers                %ld\n\n", layerMaxBlendingFraction, layerEdgeLength, layerExpansionRatio, minLayers, maxLayers);
    else OUTS("    layerMaxBlendingFraction 0 (boundary layer treatment is NOT applied)\n");

    long nPointsTot = 0, nInternalTot = 0;
    for (Rank& K : R) {
        nPointsTot += K.mesh.nPoints();
        for (uint8_t v : K.internal) nInternalTot += v;
    }
    nPointsTot = g_comm.reduceSum(nPointsTot);
    nInternalTot = g_comm.reduceSum(nInternalTot);
    OUT("Mesh includes a total of %ld points:\n  - %ld internal (non-boundary) points\n  - %ld boundary points\n", nPointsTot, nInternalTot, nPointsTot - nInternalTot);
    OUT("Mesh minimum edge length = %g\nMesh maximum edge length = %g\n\n", meshMinEdgeLength, meshMaxEdgeLength);

    for (Rank& K : R) check(smgpu_set_params(K.h, &prm), "smgpu_set_params");
    std::vector<std::vector<int64_t>> sharedGlobalOf;   // every rank's shared points (global ids, ascending): the set-up syncs
    if (opt.parallel) {
        // the copies of a point across a processorCyclic patch are not one point (their positions differ by the patch transform):
        // nothing here transforms positions or sums, so such a case is refused instead of being coupled wrongly or not at all
        for (const auto& p : K0.mesh.patches)
            if (p.type == "processorCyclic") fatal("patch " + p.name + " is a processorCyclic patch: cyclic coupling between sub-domains is not supported");
        {   // every rank must take the same route: by global ids only if ALL of them have the file
            const auto have = g_comm.allgatherVec(std::vector<int>(1, K0.pointProc.empty() ? 0 : 1));
            bool all = true;
            for (const auto& h : have) all = all && h[0];
            const auto allFaces = g_comm.allgatherVec(processorPatchFaces(K0));
            {   // either route assumes that the two sides of a processor patch pair list their faces in corresponding order
                const std::string err = checkProcessorPatchCoords(myRank, nRanks, allFaces, g_comm.allgatherVec(processorPatchCoords(K0)));
                if (!err.empty()) fatal(err);
            }
            if (!all) {
                const std::string err = labelsFromPatches(K0, myRank, nRanks, allFaces);
                if (!err.empty()) fatal(err);
                if (myRank == 0) OUT("Shared points matched through the processor patches (no pointProcAddressing)\n\n");
            }
        }
        buildHalo(K0, myRank, nRanks, sharedGroupsOf(myRank, nRanks, g_comm.allgatherVec(processorPatchLists(K0))));
        sharedGlobalOf = g_comm.allgatherVec(K0.sharedGlobal);
        for (Rank& K : R) {
            HIPCHK(hipSetDevice(K.device));
            const size_t ns = (size_t)std::max(K.nSend, 1);
            HIPCHK(hipMalloc((void**)&K.sendA, ns * SMGPU_HALO_A_DOUBLES * 8));
            HIPCHK(hipMalloc((void**)&K.sendF, ns * 4));
            HIPCHK(hipMalloc((void**)&K.localStats, 16));
            HIPCHK(hipMalloc((void**)&K.sendL, ns * SMGPU_HALO_L_DOUBLES * 8));
            if (transport == TRANSPORT_PUSH) {
                // receive buffers + flag words the peers can map (uncached device memory with an IPC handle each)
                const size_t bytes[4] = {ns * SMGPU_HALO_A_DOUBLES * 8, ns * SMGPU_HALO_L_DOUBLES * 8, ns * 4, 4 * 2 * 64};
                void* ptr[4] = {nullptr, nullptr, nullptr, nullptr};
                for (int b = 0; b < 4; ++b) check(smgpu_push_alloc(K.device, bytes[b], &ptr[b], pushHandles.h[b]), "smgpu_push_alloc");
                K.recvA = (double*)ptr[0]; K.recvL = (double*)ptr[1]; K.recvF = (int32_t*)ptr[2]; pushFlags = ptr[3];
            } else {
                HIPCHK(hipMalloc((void**)&K.recvA, ns * SMGPU_HALO_A_DOUBLES * 8));
                HIPCHK(hipMalloc((void**)&K.recvF, ns * 4));
                HIPCHK(hipMalloc((void**)&K.recvL, ns * SMGPU_HALO_L_DOUBLES * 8));
            }
            smgpu_halo_desc hd{};
            hd.nShared = (int32_t)K.sharedLocal.size(); hd.sharedLocal = K.sharedLocal.data();
            hd.nSend = K.nSend; hd.sendShared = K.sendShared.data(); hd.nRecv = K.nSend;
            hd.combOffsets = K.combOff.data(); hd.combSlots = K.combSlots.data();
            hd.sendA = K.sendA; hd.recvA = K.recvA; hd.sendF = K.sendF; hd.recvF = K.recvF; hd.localStats = K.localStats;
            hd.sendL = K.sendL; hd.recvL = K.recvL;
            check(smgpu_halo_configure(K.h, &hd), "smgpu_halo_configure");
        }
        if (transport == TRANSPORT_PUSH) {
            // every rank's handles and slot counts; the peers' buffers mapped; the layout of smoothmesh_amd/halo.py:push_layout
            // (a peer groups its receive slots by source rank, ascending: this rank's records start behind those of the
            // lower ranks; its flag word at the peer is its position among the peer's peers)
            const auto allHandles = g_comm.allgather(pushHandles);
            const auto countsOf = g_comm.allgatherVec(K0.peerCount);
            for (int o = 0; o < nRanks; ++o) {
                if (o == myRank || K0.peerCount[o] == 0) continue;
                if (countsOf[(size_t)o][(size_t)myRank] != K0.peerCount[o]) fatal("asymmetric shared-point lists");
                void* m[4] = {nullptr, nullptr, nullptr, nullptr};
                for (int b = 0; b < 4; ++b) { check(smgpu_push_open(K0.device, allHandles[(size_t)o].h[b], &m[b]), "smgpu_push_open"); pushMapped.push_back(m[b]); }
                int32_t base = 0, idx = 0;
                for (int r2 = 0; r2 < myRank; ++r2) { base += countsOf[(size_t)o][(size_t)r2]; idx += countsOf[(size_t)o][(size_t)r2] > 0 ? 1 : 0; }
                pushCount.push_back(K0.peerCount[o]); pushBase.push_back(base); pushIndex.push_back(idx);
                pushA.push_back(m[0]); pushL.push_back(m[1]); pushF.push_back(m[2]); pushFl.push_back(m[3]);
            }
            smgpu_push_desc pd{};
            pd.nPeers = (int32_t)pushCount.size();
            pd.peerCount = pushCount.data(); pd.remoteBase = pushBase.data(); pd.myIndexAtPeer = pushIndex.data();
            pd.peerRecvA = pushA.data(); pd.peerRecvL = pushL.data(); pd.peerRecvF = pushF.data(); pd.peerFlags = pushFl.data();
            pd.localFlags = pushFlags;
            g_comm.barrier();                      // every rank's flag words are zero before anybody's first store
            check(smgpu_halo_set_push(K0.h, &pd), "smgpu_halo_set_push");
        }
    }

    // -parallel: the reference's syncPointList calls of the set-ups: every rank publishes its values at its shared points and
    // combines, for each of its points, the values of the sharers in ascending rank order
    typedef int (*SharedFn)(smgpu_handle*, int32_t, int32_t, double*);
    // op 0 max, 1 sum (ascending rank), 2 larger magnitude (maxMagSqrEqOp).  The magnitude fold follows globalMeshData::syncData:
    // started from the lowest rank's value, the others folded onto it in ascending rank order, every sharer ends with that value (a
    // tie keeps the lower rank's); SMGPU_SYNC_VARIANT=own: folded onto the own value instead (include/smgpu.h smgpu_set_sync_variant)
    const bool ownFold = [] { const char* sv = std::getenv("SMGPU_SYNC_VARIANT"); return sv && std::string(sv) == "own"; }();
    auto syncShared = [&](SharedFn fn, const char* what, int field, int width, int op) {
        const size_t nS = K0.sharedGlobal.size();
        std::vector<double> mine(std::max<size_t>(nS, 1) * width, 0.0);
        if (nS) check(fn(K0.h, field, 0, mine.data()), what);
        const std::vector<std::vector<double>> sent = g_comm.allgatherVec(mine);
        std::vector<double> v(mine);
        for (size_t i = 0; i < nS; ++i) {
            const int64_t g = K0.sharedGlobal[i];
            double* x = &v[i * width];
            if (op == 1) for (int c = 0; c < width; ++c) x[c] = 0.0;
            bool first = true;
            for (int o : K0.sharersOf[i]) {                              // the point's group (not every rank that holds the id)
                const double* y = nullptr;
                if (o == myRank) y = &mine[i * width];
                else {
                    const auto& sg = sharedGlobalOf[(size_t)o];
                    const auto it = std::lower_bound(sg.begin(), sg.end(), g);
                    if (it == sg.end() || *it != g) fatal("shared-point tables disagree between ranks");
                    y = &sent[(size_t)o][(size_t)(it - sg.begin()) * width];
                }
                if (op == 1) { for (int c = 0; c < width; ++c) x[c] = x[c] + y[c]; continue; }
                if (op == 0) { if (o != myRank && y[0] > x[0]) x[0] = y[0]; continue; }
                if (ownFold) { if (o == myRank) continue; }
                else if (first) { x[0] = y[0]; x[1] = y[1]; x[2] = y[2]; first = false; continue; }   // the master's value
                const double mx = x[0] * x[0] + x[1] * x[1] + x[2] * x[2], my = y[0] * y[0] + y[1] * y[1] + y[2] * y[2];
                if (!(mx >= my)) { x[0] = y[0]; x[1] = y[1]; x[2] = y[2]; }
            }
        }
        if (nS) check(fn(K0.h, field, 1, v.data()), what);
    };

    if (doLayerTreatment) {   // set-up SM.C:2215-2221 on the engines' side
        std::vector<std::vector<int32_t>> pStart(R.size()), pSize(R.size());
        std::vector<std::vector<uint8_t>> pKind(R.size());
        std::vector<smgpu_layer_desc> ld(R.size());
        for (size_t r = 0; r < R.size(); ++r) {
            for (const PatchInfo& p : R[r].mesh.patches) {
                pStart[r].push_back(p.startFace);
                pSize[r].push_back(p.nFaces);
                pKind[r].push_back(p.type == "processor" ? 1 : (p.type == "empty" ? 2 : 0));
            }
            ld[r] = smgpu_layer_desc{};
            ld[r].nPatches = (int32_t)pStart[r].size();
            ld[r].patchStart = pStart[r].data(); ld[r].patchSize = pSize[r].data(); ld[r].patchKind = pKind[r].data();
            ld[r].isLayerPatch = isLayerPatchOf[r].data();
            ld[r].layerMaxBlendingFraction = layerMaxBlendingFraction; ld[r].layerEdgeLength = layerEdgeLength;
            ld[r].layerExpansionRatio = layerExpansionRatio; ld[r].minLayers = (int32_t)minLayers; ld[r].maxLayers = (int32_t)maxLayers;
        }
        int32_t on = 0, maxIter = 0;
        if (!opt.parallel) check(smgpu_set_layers(R[0].h, &ld[0], &on), "smgpu_set_layers");
        else {
            auto sync = [&](int field, int width, int op) { syncShared(smgpu_layers_shared, "smgpu_layers_shared", field, width, op); };
            check(smgpu_layers_begin(K0.h, &ld[0], &on, &maxIter), "smgpu_layers_begin");
            for (int it = 0; it < maxIter; ++it) {
                for (Rank& K : R) check(smgpu_layers_step(K.h, SMGPU_LAYERS_HOPS_SWEEP, 0), "smgpu_layers_step");
                sync(SMGPU_LAYERS_F_HOPS, 1, 0);                          // OBB.C:124-130
            }
            for (Rank& K : R) check(smgpu_layers_step(K.h, SMGPU_LAYERS_NORMALS_ACCUMULATE, 0), "smgpu_layers_step");
            sync(SMGPU_LAYERS_F_NORMALS_COUNT, 4, 1);                     // OBB.C:184-198
            for (Rank& K : R) check(smgpu_layers_step(K.h, SMGPU_LAYERS_NORMALS_FINISH, 0), "smgpu_layers_step");
            for (int it = 1; it <= maxIter; ++it) {
                for (Rank& K : R) check(smgpu_layers_step(K.h, SMGPU_LAYERS_PROPAGATE_SWEEP, it), "smgpu_layers_step");
                sync(SMGPU_LAYERS_F_NORMALS, 3, 2);                       // OBB.C:359-365
            }
            for (Rank& K : R) check(smgpu_layers_step(K.h, SMGPU_LAYERS_FINISH, 0), "smgpu_layers_step");
        }
    }

    if (doBoundarySmoothing) {   // SM.C:2131-2172, 2186-2253 on the engine's side
        std::vector<double> surfPts, initPts, tgtPts;
        std::vector<int32_t> surfTris, initE, tgtE;
        try {
            readObjSurface(cd + "/" + targetSurfacesFile, surfPts, surfTris);
            OUT("Target surfaces file \"%s\" stats:\nTriangles    : %zu\nVertices     : %zu\n\n", targetSurfacesFile.c_str(), surfTris.size() / 3, surfPts.size() / 3);
            if (fileExists(cd + "/" + initEdgesFile)) {
                readObjEdges(cd + "/" + initEdgesFile, initPts, initE);
                OUT("Initial feature edges file \"%s\" stats:\n  points : %zu\n  edges  : %zu\n\n", initEdgesFile.c_str(), initPts.size() / 3, initE.size() / 2);
            }
            if (fileExists(cd + "/" + targetEdgesFile)) {
                readObjEdges(cd + "/" + targetEdgesFile, tgtPts, tgtE);
                OUT("Target feature edges file \"%s\" stats:\n  points : %zu\n  edges  : %zu\n", targetEdgesFile.c_str(), tgtPts.size() / 3, tgtE.size() / 2);
            } else
                OUT("WARNING: Initial feature edges will be used also as target edges, because\ndid not find file %s.\n\n", targetEdgesFile.c_str());
        } catch (const std::exception& e) { fatal(e.what()); }
        std::vector<std::vector<int32_t>> pStart(R.size()), pSize(R.size());
        std::vector<std::vector<uint8_t>> pKind(R.size());
        std::vector<smgpu_boundary_desc> bd(R.size());
        for (size_t r = 0; r < R.size(); ++r) {
            for (const auto& pt : R[r].mesh.patches) {
                pStart[r].push_back(pt.startFace); pSize[r].push_back(pt.nFaces);
                pKind[r].push_back(pt.type == "processor" ? 1 : pt.type == "empty" ? 2 : 0);
            }
            smgpu_boundary_desc& d = bd[r];
            d = smgpu_boundary_desc{};
            d.nPatches = (int32_t)R[r].mesh.patches.size(); d.patchStart = pStart[r].data(); d.patchSize = pSize[r].data(); d.patchKind = pKind[r].data();
            d.isSmoothingPatch = isSmoothingPatchOf[r].data();
            d.nInitEdgePoints = (int32_t)(initPts.size() / 3); d.initEdgePoints = initPts.data(); d.nInitEdges = (int32_t)(initE.size() / 2); d.initEdges = initE.data();
            d.nTargetEdgePoints = (int32_t)(tgtPts.size() / 3); d.targetEdgePoints = tgtPts.data(); d.nTargetEdges = (int32_t)(tgtE.size() / 2); d.targetEdges = tgtE.data();
            d.nSurfacePoints = (int32_t)(surfPts.size() / 3); d.surfacePoints = surfPts.data();
            d.nSurfaceTriangles = (int32_t)(surfTris.size() / 3); d.surfaceTriangles = surfTris.data();
            d.isCornerPointIO = isCornerPointIO[r].empty() ? nullptr : isCornerPointIO[r].data();
            d.isFeatureEdgePointIO = isFeatureEdgePointIO[r].empty() ? nullptr : isFeatureEdgePointIO[r].data();
            d.distanceTolerance = 1e-4 * std::min(meshMinEdgeLength, layerEdgeLength);   // REL_TOL, SM.C:1921
            d.internalSmoothingBlendingFraction = internalSmoothingBlendingFraction;
        }
        OUT("Distance tolerance = %g\n\n", bd[0].distanceTolerance);
        smgpu_boundary_info tot{};
        if (!opt.parallel) {
            check(smgpu_set_boundary_smoothing(R[0].h, &bd[0], &tot), "smgpu_set_boundary_smoothing");
        } else {
            // the reductions of getMeshStats (SM.C:1528-1538), then the set-up in steps with its syncPointList calls
            double mn = 1e300, bb[6] = {1e300, -1e300, 1e300, -1e300, 1e300, -1e300};
            for (Rank& K : R) {
                double m1, b1[6];
                check(smgpu_boundary_stats(K.h, &m1, b1), "smgpu_boundary_stats");
                mn = std::min(mn, m1);
                for (int c = 0; c < 6; c += 2) { bb[c] = std::min(bb[c], b1[c]); bb[c + 1] = std::max(bb[c + 1], b1[c + 1]); }
            }
            mn = g_comm.reduceMin(mn);                                         // returnReduce, SM.C:1528-1535
            for (int c = 0; c < 6; c += 2) { bb[c] = g_comm.reduceMin(bb[c]); bb[c + 1] = g_comm.reduceMax(bb[c + 1]); }
            const double perimeter = bb[1] - bb[0] + bb[3] - bb[2] + bb[5] + bb[4];
            {
                smgpu_boundary_info bi{};
                check(smgpu_boundary_begin(K0.h, &bd[0], mn, perimeter, &bi), "smgpu_boundary_begin");
                tot = bi;   // returnReduce sumOp, BPS.C:423-427
                tot.enabled = g_comm.reduceAnd(bi.enabled != 0) ? 1 : 0;
                tot.nCornerPoints = (int32_t)g_comm.reduceSum(bi.nCornerPoints); tot.nFeatureEdgePoints = (int32_t)g_comm.reduceSum(bi.nFeatureEdgePoints);
                tot.nSmoothingSurfacePoints = (int32_t)g_comm.reduceSum(bi.nSmoothingSurfacePoints);
                tot.nFrozenSurfacePoints = (int32_t)g_comm.reduceSum(bi.nFrozenSurfacePoints);
            }
            if (tot.enabled) {
                for (int it = 0; it < 2; ++it) {   // SM.C:2218
                    for (Rank& K : R) check(smgpu_boundary_step(K.h, SMGPU_BOUNDARY_HOPS_SWEEP), "smgpu_boundary_step");
                    syncShared(smgpu_boundary_shared, "smgpu_boundary_shared", SMGPU_BOUNDARY_F_HOPS, 1, 0);   // OBB.C:124-130
                }
                for (Rank& K : R) check(smgpu_boundary_step(K.h, SMGPU_BOUNDARY_TABLES), "smgpu_boundary_step");
                for (Rank& K : R) check(smgpu_boundary_step(K.h, SMGPU_BOUNDARY_NORMALS_ACCUMULATE), "smgpu_boundary_step");
                syncShared(smgpu_boundary_shared, "smgpu_boundary_shared", SMGPU_BOUNDARY_F_NORMALS_COUNT, 4, 1);   // OBB.C:184-198
                for (Rank& K : R) check(smgpu_boundary_step(K.h, SMGPU_BOUNDARY_NORMALS_FINISH), "smgpu_boundary_step");
            }
        }
        if (!tot.enabled) fatal("boundary point smoothing: the engine did not enable it (empty target surface or edge mesh?)");
        OUT("Detected number of target edge mesh strings: %d\n\n", tot.nTargetEdgeStrings);
        OUT("Boundary point classification summary:\n- Detected number of corner points: %d\n- Detected number of feature edge points: %d\n"
                    "- Detected number of smoothing surface points: %d\n- Detected number of frozen surface points: %d\n\n",
                    tot.nCornerPoints, tot.nFeatureEdgePoints, tot.nSmoothingSurfacePoints, tot.nFrozenSurfacePoints);
    }

    int32_t lDoubles = SMGPU_HALO_L_LAYERS;   // doubles per slot of the L records in use (all engines agree)
    if (opt.parallel) check(smgpu_halo_l_doubles(K0.h, &lDoubles), "smgpu_halo_l_doubles");
    std::vector<int> peerBaseOf((size_t)nRanks, 0);   // first slot rank o keeps towards me
    std::vector<int> nSendOf((size_t)nRanks, 0);      // every rank's number of send slots (the part bases in its slot of the mapping)
    if (opt.parallel) {
        const std::vector<std::vector<int>> allBase = g_comm.allgatherVec(K0.peerSendBase);
        for (int o = 0; o < nRanks; ++o) peerBaseOf[(size_t)o] = allBase[(size_t)o][(size_t)myRank];
        nSendOf = g_comm.allgather(K0.nSend);
    }
    const bool withL = doLayerTreatment || doBoundarySmoothing;
    // One exchange = for every rank that shares points with this one, its slots of the send buffer against the matching slots of
    // the receive buffer.  RCCL: one group of ncclSend / ncclRecv pairs on the engine's stream (exchange A carries the L records
    // in the same group: one collective kernel); nothing waits on the host.  shm: the same records staged through the ranks'
    // mapping (D2H, barrier, H2D, barrier).
    struct Part { const void* send; void* recv; size_t bytesPerSlot; };
    // SMOOTHMESH_EXCHANGE_STREAM=1 (RCCL transport, opt-in until it has run on a node with several GPUs -- scripts/first_multi_gpu.sh):
    // the send / recv groups go onto a high-priority stream of their own and the engine orders them against its kernels with flag
    // words (include/smgpu.h, smgpu_halo_set_exchange_stream: with the constraints off the two multi-role launches then run next
    // to both exchanges -- one rank of eight: 123 instead of 135 us per iteration in the Python driver's probe).  Set behind the
    // start-up self-check below, which uses the engine's stream.
    hipStream_t xchgStream = engineStream;
    auto enableExchangeStream = [&]() {
        const char* v = std::getenv("SMOOTHMESH_EXCHANGE_STREAM");
        if (!(v && std::atoi(v) != 0) || transport != TRANSPORT_RCCL || !opt.parallel) return;
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));      // (hi = the numerically lowest value = the highest priority)
        HIPCHK(hipStreamCreateWithPriority(&xchgStream, hipStreamNonBlocking, hi));
        check(smgpu_halo_set_exchange_stream(K0.h, 1, (void*)xchgStream), "smgpu_halo_set_exchange_stream");
        if (myRank == 0) OUT("Exchanges on a stream of their own (SMOOTHMESH_EXCHANGE_STREAM)\n\n");
    };
    auto syncExchangeStream = [&]() { if (xchgStream != engineStream) HIPCHK(hipStreamSynchronize(xchgStream)); };
    auto exchange = [&](std::initializer_list<Part> parts) {
        if (transport == TRANSPORT_PUSH) return;   // the pack kernels have stored the records at the peers themselves
        if (transport == TRANSPORT_RCCL) {
            NCCLCHK(ncclGroupStart());
            for (const Part& pt : parts)
                for (int o = 0; o < nRanks; ++o) {
                    const size_t c = (size_t)K0.peerCount[o];
                    if (!c) continue;
                    const size_t off = (size_t)K0.peerSendBase[o] * pt.bytesPerSlot;
                    NCCLCHK(ncclSend((const char*)pt.send + off, c * pt.bytesPerSlot, ncclChar, o, nccl, xchgStream));
                    NCCLCHK(ncclRecv((char*)pt.recv + off, c * pt.bytesPerSlot, ncclChar, o, nccl, xchgStream));
                }
            NCCLCHK(ncclGroupEnd());
            return;
        }
        // debug transport: my slot of the mapping = [part 0 send buffer | part 1 send buffer | ...]
        size_t off = 0;
        std::vector<size_t> base;
        for (const Part& pt : parts) {
            base.push_back(off);
            const size_t bytes = (size_t)K0.nSend * pt.bytesPerSlot;
            if (off + bytes > g_comm.slotBytes()) fatal("shm transport: send buffers exceed the mapping");
            if (bytes) HIPCHK(hipMemcpyAsync(g_comm.slot(myRank) + off, pt.send, bytes, hipMemcpyDeviceToHost, engineStream));
            off += (bytes + 63) & ~(size_t)63;
        }
        HIPCHK(hipStreamSynchronize(engineStream));
        g_comm.barrier();                                                 // every rank's records are in its slot
        size_t pi = 0;
        for (const Part& pt : parts) {
            for (int o = 0; o < nRanks; ++o) {
                const size_t c = (size_t)K0.peerCount[o];
                if (!c) continue;
                size_t obase = 0;                                         // start of this part in rank o's slot
                { size_t q = 0; for (const Part& pp : parts) { if (q == pi) break; obase += (((size_t)nSendOf[(size_t)o] * pp.bytesPerSlot) + 63) & ~(size_t)63; ++q; } }
                const size_t theirOff = (size_t)peerBaseOf[(size_t)o] * pt.bytesPerSlot;   // where rank o keeps its slots towards me
                HIPCHK(hipMemcpyAsync((char*)pt.recv + (size_t)K0.peerSendBase[o] * pt.bytesPerSlot, g_comm.slot(o) + obase + theirOff, c * pt.bytesPerSlot,
                                      hipMemcpyHostToDevice, engineStream));
            }
            ++pi;
        }
        HIPCHK(hipStreamSynchronize(engineStream));
        g_comm.barrier();                                                 // nobody overwrites its slot before everyone has read it
    };

    // One-shot check of the RCCL transport before the first iteration (the Python driver has the same, rccl_direct.py): every
    // rank sends a rank-specific pattern through the very exchange the loop uses and compares what arrived with what the peers
    // must have sent (peer o keeps its slots towards this rank from peerBaseOf[o] on).  A layout or ordering mistake in the
    // send / recv groups would otherwise produce wrong meshes silently on the first real multi-GPU run.
    if (opt.parallel && transport == TRANSPORT_RCCL && nRanks > 1) {
        const int32_t kMul = 1000003;
        std::vector<int32_t> pat((size_t)std::max(K0.nSend, 1)), got((size_t)std::max(K0.nSend, 1), -1);
        for (int k = 0; k < K0.nSend; ++k) pat[(size_t)k] = myRank * kMul + k;
        if (K0.nSend) {
            HIPCHK(hipMemcpyAsync(K0.sendF, pat.data(), (size_t)K0.nSend * 4, hipMemcpyHostToDevice, engineStream));
            HIPCHK(hipMemsetAsync(K0.recvF, 0xff, (size_t)K0.nSend * 4, engineStream));
        }
        exchange({Part{K0.sendF, K0.recvF, 4}});
        if (K0.nSend) HIPCHK(hipMemcpyAsync(got.data(), K0.recvF, (size_t)K0.nSend * 4, hipMemcpyDeviceToHost, engineStream));
        HIPCHK(hipStreamSynchronize(engineStream));
        bool good = true;
        for (int o = 0; o < nRanks; ++o)
            for (int j = 0; j < K0.peerCount[o]; ++j)
                good = good && got[(size_t)(K0.peerSendBase[o] + j)] == o * kMul + peerBaseOf[(size_t)o] + j;
        if (g_comm.reduceAnd(good)) OUTS("RCCL exchange self-check: pass\n");
        else
            fatal("RCCL exchange self-check failed: the records that arrived are not the ones the peers sent (set SMOOTHMESH_TRANSPORT=shm to run on the host-staged transport)");
    }
    enableExchangeStream();

    auto writeMesh = [&](double timeValue) {
        const auto tw = std::chrono::steady_clock::now();
        struct Stop { double& acc; std::chrono::steady_clock::time_point a; ~Stop() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); } } stop{tWrite, tw};
        const std::string tn = timeName(timeValue);
        OUT("Writing new mesh to time %s\n\n", tn.c_str());
        for (Rank& K : R) {
            std::vector<double> pts((size_t)K.mesh.nPoints() * 3);
            check(smgpu_get_points(K.h, pts.data()), "smgpu_get_points");
            try { writePoints(K.root + "/" + tn + "/polyMesh", tn + "/polyMesh", K.mesh.nPoints(), pts.data(), binary, writePrecision); }
            catch (const std::exception& e) { fatal(e.what()); }
        }
        if (doBoundarySmoothing)   // labelIOLists with AUTO_WRITE, SM.C:2039-2064 (per sub-domain under -parallel)
            for (Rank& K : R) {
                std::vector<int32_t> a((size_t)K.mesh.nPoints()), b((size_t)K.mesh.nPoints());
                check(smgpu_get_boundary_classification(K.h, a.data(), b.data()), "smgpu_get_boundary_classification");
                try {
                    writeLabelList(K.root + "/" + tn + "/isCornerPoint", tn, "isCornerPoint", "labelList", (int64_t)a.size(), a.data(), binary, "");
                    writeLabelList(K.root + "/" + tn + "/isFeatureEdgePoint", tn, "isFeatureEdgePoint", "labelList", (int64_t)b.size(), b.data(), binary, "");
                } catch (const std::exception& e) { fatal(e.what()); }
            }
    };

    // the loop, SM.C:2257-2437
    bool stopIteration = false;
    long i = 0;
    double timeValue = startIsConstant ? 0.0 : startValue;
    std::vector<smgpu_iter_stats> stats;
    double* dHist = nullptr;       // -parallel, relTol <= 0: the chunk's {residual, nFrozenPoints} records on the device
    size_t histCap = 0;
    while (i < centroidalIters && !stopIteration) {
        // run up to the next write point in one engine call (no host synchronisation inside)
        long chunk = std::min(centroidalIters - i, writeInterval - (i % writeInterval));
        int32_t done = 0;
        stats.assign((size_t)chunk, smgpu_iter_stats{});
        const auto tl = std::chrono::steady_clock::now();
        if (!opt.parallel) {
            check(smgpu_iterate(R[0].h, (int32_t)chunk, relTol, stats.data(), &done), "smgpu_iterate");
        } else {
            // relTol <= 0 cannot stop the loop (residual >= 0, SM.C:2401): the engine then keeps the chunk's per-iteration
            // {residual, nFrozenPoints} on the device and the host reads nothing until the chunk is over
            const bool noStop = !(relTol > 0.0);
            if (noStop) {
                if ((long)histCap < chunk) {
                    if (dHist) HIPCHK(hipFree(dHist));
                    HIPCHK(hipMalloc((void**)&dHist, (size_t)chunk * 16));
                    histCap = (size_t)chunk;
                }
                check(smgpu_halo_set_stats_history(K0.h, dHist, (int32_t)chunk), "smgpu_halo_set_stats_history");
            }
            // peer stores: a consuming kernel waits for its peers' flags for a BOUNDED time (a dead rank must not hang the others), so
            // the ranks enter a chunk together -- one of them may still have been writing its sub-domain, or reading a case, for
            // seconds (the RCCL and host-staged transports simply wait there)
            if (transport == TRANSPORT_PUSH) g_comm.barrier();
            for (long k = 0; k < chunk; ++k) {
                check(smgpu_iter_begin(K0.h), "smgpu_iter_begin");
                if (withL) exchange({Part{K0.sendA, K0.recvA, SMGPU_HALO_A_DOUBLES * 8}, Part{K0.sendL, K0.recvL, (size_t)lDoubles * 8}});   // SM.C:134-148, 402-478; OBB.C:184-198, 490-496
                else exchange({Part{K0.sendA, K0.recvA, SMGPU_HALO_A_DOUBLES * 8}});
                check(smgpu_iter_mid(K0.h), "smgpu_iter_mid");
                exchange({Part{K0.sendF, K0.recvF, 4}});                                                   // SM.C:2374
                check(smgpu_iter_end(K0.h), "smgpu_iter_end");
                ++done;
                if (noStop) continue;
                double ls[2];
                HIPCHK(hipMemcpyAsync(ls, K0.localStats, 16, hipMemcpyDeviceToHost, engineStream));
                HIPCHK(hipStreamSynchronize(engineStream));
                const double res = g_comm.reduceMax(ls[0]);                      // returnReduce maxOp, SM.C:1567
                const long nf = g_comm.reduceSum((long)ls[1]);                   // returnReduce sumOp, SM.C:2396
                stats[(size_t)k].residual = res;
                stats[(size_t)k].nFrozenPoints = (int32_t)nf;
                if (res < relTol) break;
            }
            // an error word raised by a kernel of this chunk (a peer's records that never came, a grid barrier that could not
            // complete, a point without usable neighbours): found here, not a writeInterval later
            check(smgpu_check_error(K0.h), "smgpu_check_error");
            syncExchangeStream();
            if (noStop) {
                check(smgpu_halo_set_stats_history(K0.h, nullptr, 0), "smgpu_halo_set_stats_history");   // closes the last iteration
                std::vector<double> hist((size_t)chunk * 2);
                HIPCHK(hipMemcpyAsync(hist.data(), dHist, hist.size() * 8, hipMemcpyDeviceToHost, engineStream));
                HIPCHK(hipStreamSynchronize(engineStream));
                const std::vector<std::vector<double>> all = g_comm.allgatherVec(hist);
                for (long k = 0; k < chunk; ++k) {
                    double res = 0.0, nf = 0.0;
                    for (const auto& hh : all) { res = std::max(res, hh[(size_t)2 * k]); nf += hh[(size_t)2 * k + 1]; }
                    stats[(size_t)k].residual = res;
                    stats[(size_t)k].nFrozenPoints = (int32_t)nf;
                }
            }
        }
        tLoop += secondsSince(tl);
        mark("a stretch of the loop done");
        for (int32_t k = 0; k < done; ++k)
            OUT("Smoothing iteration=%ld nFrozenPoints=%d residual=%g\n", i + k + 1, stats[(size_t)k].nFrozenPoints, stats[(size_t)k].residual);
        i += done;
        timeValue += done * deltaT;   // runTime++ per iteration, SM.C:2414
        const bool hitTol = done > 0 && stats[(size_t)done - 1].residual < relTol;
        if (hitTol) { OUTS("Residual reached relTol, stopping."); stopIteration = true; }
        if (i == centroidalIters) { OUTS("Maximum centroidalIters reached, stopping."); stopIteration = true; }
        // SM.C:2416: write at stop or every writeInterval iterations (not after the very first one)
        if (stopIteration || ((i % writeInterval) == 0 && i > 1)) writeMesh(timeValue);
        if (done == 0) break;
    }

    // Near-tie census (include/smgpu.h): the engine's acos may differ from the reference's in the last bit, so only an angle
    // comparison with sides a few ulp apart could have been decided the other way by the reference -- say so when there was one
    {
        int64_t nt[4] = {0, 0, 0, 0};
        check(smgpu_get_near_ties(K0.h, nt), "smgpu_get_near_ties");
        const long tot = g_comm.reduceSum((long)nt[0]);
        if (tot > 0)
            OUT("WARNING: %ld angle comparison(s) of this run had their two sides within 4 ulp of each other (edge angle %lld, face-angle range %lld, "
                "face-angle walk %lld on the master): the reference's acos may decide such a comparison the other way "
                "(expected where points move by a few ulp only, i.e. in converged regions)\n",
                tot, (long long)nt[1], (long long)nt[2], (long long)nt[3]);
    }
    if (nccl) { HIPCHK(hipStreamSynchronize(engineStream)); syncExchangeStream(); NCCLCHK(ncclCommDestroy(nccl)); }
    if (transport == TRANSPORT_PUSH && opt.parallel) {
        HIPCHK(hipStreamSynchronize(engineStream));
        g_comm.barrier();                          // nobody unmaps a buffer a peer may still store into
        check(smgpu_halo_set_push(K0.h, nullptr), "smgpu_halo_set_push");
        for (void* m : pushMapped) check(smgpu_push_close(m), "smgpu_push_close");
        g_comm.barrier();
    }
    mark("loop and output done");
    for (Rank& K : R) smgpu_destroy(K.h);
    mark("engine destroyed");
    g_comm.barrier();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    OUT("ClockTime breakdown: read %.2f s, engine set-up %.2f s, smoothing loop %.3f s, write %.2f s, other %.2f s\n", tRead, tCreate, tLoop, tWrite,
        std::max(0.0, secs - tRead - tCreate - tLoop - tWrite));
    OUT("ClockTime = %d s.\n\nEnd\n", (int)secs);
    return 0;
}
