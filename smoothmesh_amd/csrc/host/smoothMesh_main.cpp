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
// -parallel: the case holds processorN/ sub-domains (decomposePar layout, with pointProcAddressing);
// this single process drives one engine per sub-domain, placed round-robin on the visible GPUs, and
// moves the shared-point buffers with device-to-device copies.  (The measured multi-GPU path is the
// one-process-per-GPU RCCL driver in smoothmesh_amd/halo.py; this one keeps the CLI usable on any
// number of GPUs.)
#include <dirent.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <regex>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/smgpu.h"
#include "polymesh_io.hpp"

using namespace smhost;

namespace {

const double VSMALL = 1.0e-300, SMALL = 1.0e-15;

[[noreturn]] void fatal(const std::string& msg) {   // FatalError << ... << abort(FatalError)
    std::fprintf(stdout, "\n\n--> FOAM FATAL ERROR: \n%s\n\nFOAM aborting\n\n", msg.c_str());
    std::fflush(stdout);
    std::exit(1);
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

struct Rank {
    std::string root;            // case dir (serial) or processorN dir
    PolyMeshData mesh;
    std::vector<uint8_t> internal;
    std::vector<int32_t> pointProc;
    smgpu_handle* h = nullptr;
    int device = 0;
    // halo
    std::vector<int32_t> sharedLocal, sendShared, combOff, combSlots;
    std::vector<int64_t> sharedGlobal;
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

void buildHalo(std::vector<Rank>& R) {
    const int n = (int)R.size();
    std::vector<std::vector<int64_t>> cand(n);
    for (int r = 0; r < n; ++r) {
        std::set<int64_t> s;
        const auto& m = R[r].mesh;
        for (const auto& p : m.patches)
            if (p.type == "processor")
                for (int32_t f = p.startFace; f < p.startFace + p.nFaces; ++f)
                    for (int32_t k = m.faceOffsets[f]; k < m.faceOffsets[f + 1]; ++k) s.insert(R[r].pointProc[(size_t)m.facePoints[k]]);
        cand[r].assign(s.begin(), s.end());
    }
    for (int r = 0; r < n; ++r) {
        Rank& K = R[r];
        std::vector<std::vector<int64_t>> shared(n);
        std::set<int64_t> all;
        for (int o = 0; o < n; ++o) {
            if (o == r) continue;
            std::set_intersection(cand[r].begin(), cand[r].end(), cand[o].begin(), cand[o].end(), std::back_inserter(shared[o]));
            all.insert(shared[o].begin(), shared[o].end());
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
}

}  // namespace

int main(int argc, char** argv) {
    const auto t0 = std::chrono::steady_clock::now();
    const Options opt = parseArgs(argc, argv);
    const std::string& cd = opt.caseDir;

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

    // sub-domain roots
    std::vector<Rank> R;
    if (opt.parallel) {
        for (int r = 0;; ++r) {
            const std::string root = cd + "/processor" + std::to_string(r);
            if (!dirExists(root)) break;
            R.emplace_back();
            R.back().root = root;
        }
        if (R.empty()) fatal("-parallel: no processor0 directory in " + cd + " (run decomposePar first)");
    } else {
        R.emplace_back();
        R.back().root = cd;
    }
    const int nRanks = (int)R.size();

    // start time (SM.C:1791-1803; controlDict startFrom latestTime)
    const auto times = listTimes(R[0].root);
    bool startIsConstant = false;
    double startValue = 0.0;
    if (opt.found("time")) {
        if (opt.kv.at("time") == "constant") startIsConstant = true;
        else startValue = opt.getD("time", 0.0);
    } else if (!times.empty()) startValue = times.back().first;
    else startIsConstant = true;

    std::printf("smoothMesh (MI355X engine %s)\nCase: %s%s\n", smgpu_version(), cd.c_str(), opt.parallel ? "  [parallel]" : "");
    std::printf("Create mesh for time = %s\n\n", startIsConstant ? "constant" : timeName(startValue).c_str());

    try {
        for (Rank& K : R) {
            const std::string meshDir = findInstance(K.root, listTimes(K.root), startValue, startIsConstant, "faces");
            const std::string ptsDir = findInstance(K.root, listTimes(K.root), startValue, startIsConstant, "points");
            readPolyMesh(meshDir, ptsDir == meshDir ? "" : ptsDir, K.mesh);
            K.internal = findInternalMeshPoints(K.mesh);
            if (opt.parallel) {
                const std::string ppa = K.root + "/constant/polyMesh/pointProcAddressing";
                if (!fileExists(ppa) && !fileExists(ppa + ".gz")) fatal(ppa + " not found (written by decomposePar; needed to match shared points)");
                readLabelList(ppa, K.pointProc);
                if ((int32_t)K.pointProc.size() != K.mesh.nPoints()) fatal(ppa + ": size does not match the number of points");
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
    if (anyLayerPatch) std::printf("Patches for boundary layer treatment: %s\n", opt.kv.at("layerPatches").c_str());
    else std::puts("Patches for boundary layer treatment: none");
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
    if (anySmoothingPatch) std::printf("Patches for boundary point smoothing: %s\n", smoothingOpt.c_str());
    else std::puts("Patches for boundary point smoothing: none");
    const double internalSmoothingBlendingFraction = opt.getD("internalSmoothingBlendingFraction", 0.0);   // SM.C:1907

    if (doLayerTreatment) std::puts("Enabled boundary layer treatment\n");
    else std::puts("Boundary layer treatment is disabled. Either no layerPatches were specified or boundaryMaxBlendingFraction is zero\n");

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
    if (labelIOListsHaveData) std::puts("Found corners and feature edges in isCornerPoint and isFeatureEdgePoint files\n");
    else std::puts("Did not find corners and feature edges in isCornerPoint and isFeatureEdgePoint files\n");

    // prerequisites of the boundary point smoothing, SM.C:2080-2093
    const std::string targetSurfacesFile = "constant/geometry/targetSurfaces.obj", initEdgesFile = "constant/geometry/initEdges.obj",
                      targetEdgesFile = "constant/geometry/targetEdges.obj";
    const bool doBoundarySmoothing = fileExists(cd + "/" + targetSurfacesFile) && (fileExists(cd + "/" + initEdgesFile) || labelIOListsHaveData) &&
                                     anySmoothingPatch;
    if (doBoundarySmoothing) std::puts("Enabled boundary point smoothing\n");
    else std::printf("Boundary point smoothing is disabled. Missing smoothingPatches, or one or both of files:\n%s\n%s\n\n", targetSurfacesFile.c_str(), initEdgesFile.c_str());
    if (doLayerTreatment && !doBoundarySmoothing)   // SM.C:2095-2098
        std::puts("WARNING: Boundary layer treatment will be done without boundary point smoothing. This can result in distorted boundary cells.\n");

    // engines
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0) fatal("no HIP device available (this build has no CPU fallback)");
    const int dev0 = (int)opt.getL("device", 0);
    for (int r = 0; r < nRanks; ++r) {
        Rank& K = R[r];
        smgpu_mesh_desc d{};
        d.nPoints = K.mesh.nPoints(); d.nCells = K.mesh.nCells; d.nFaces = K.mesh.nFaces(); d.nInternalFaces = K.mesh.nInternalFaces();
        d.points = K.mesh.points.data(); d.faceOffsets = K.mesh.faceOffsets.data(); d.facePoints = K.mesh.facePoints.data();
        d.owner = K.mesh.owner.data(); d.neighbour = K.mesh.neighbour.data();
        d.isInternalPoint = K.internal.data(); d.isSmoothingSurfacePoint = nullptr;
        K.device = (dev0 + r) % nDev;
        d.device = K.device; d.stream = nullptr; d.useCallerStream = 0;
        check(smgpu_create(&d, &K.h), "smgpu_create");
    }

    // getMeshStats + defaults (SM.C:1857-1918)
    double meshMinEdgeLength = 1e300, meshMaxEdgeLength = 0.0;
    for (Rank& K : R) {
        double a, b;
        check(smgpu_mesh_stats(K.h, &a, &b), "smgpu_mesh_stats");
        meshMinEdgeLength = std::min(meshMinEdgeLength, a);
        meshMaxEdgeLength = std::max(meshMaxEdgeLength, b);
    }
    smgpu_params prm{};
    prm.minEdgeLength = opt.getD("minEdgeLength", 0.5 * meshMinEdgeLength);
    prm.maxStepLength = opt.getD("maxStepLength", 0.3 * prm.minEdgeLength);
    if (prm.maxStepLength > 0.5 * prm.minEdgeLength)
        std::puts("WARNING: The maximum allowed step length is more than half of the minimum edge length! This may cause unstability in smoothing.\n");
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
    std::puts("Applying following parameter values in smoothing:");
    std::printf("    centroidalIters        %ld\n    relTol                 %g\n    minEdgeLength          %g\n", centroidalIters, relTol, prm.minEdgeLength);
    std::printf("    maxStepLength          %g\n    relStepFrac            %g\n    totalMinFreeze         %d\n", prm.maxStepLength, prm.relStepFrac, prm.totalMinFreeze);
    if (prm.edgeAngleConstraint) std::printf("    edgeAngleConstraint    true\n    minAngle               %g\n", prm.minAngle);
    else std::puts("    edgeAngleConstraint    false (edge min angle quality constraint is NOT applied)");
    if (prm.faceAngleConstraint) std::printf("    faceAngleConstraint    true\n    minAngle               %g\n    maxAngle               %g\n", prm.minAngle, prm.maxAngle);
    else std::puts("    faceAngleConstraint    false (face angle quality constraints are NOT applied)");
    const double layerEdgeLength = opt.getD("layerEdgeLength", prm.minEdgeLength);       // SM.C:1895-1905
    const double layerExpansionRatio = opt.getD("layerExpansionRatio", 1.3);
    const long minLayers = opt.getL("minLayers", 1), maxLayers = opt.getL("maxLayers", 4);
    if (layerMaxBlendingFraction > SMALL)
        std::printf("    layerMaxBlendingFraction %g\n    layerEdgeLength          %g\n    layerExpansionRatio      %g\n    minLayers                %ld\n"
                    "    maxLayers                %ld\n\n", layerMaxBlendingFraction, layerEdgeLength, layerExpansionRatio, minLayers, maxLayers);
    else std::puts("    layerMaxBlendingFraction 0 (boundary layer treatment is NOT applied)\n");

    long nPointsTot = 0, nInternalTot = 0;
    for (Rank& K : R) {
        nPointsTot += K.mesh.nPoints();
        for (uint8_t v : K.internal) nInternalTot += v;
    }
    std::printf("Mesh includes a total of %ld points:\n  - %ld internal (non-boundary) points\n  - %ld boundary points\n", nPointsTot, nInternalTot, nPointsTot - nInternalTot);
    std::printf("Mesh minimum edge length = %g\nMesh maximum edge length = %g\n\n", meshMinEdgeLength, meshMaxEdgeLength);

    for (Rank& K : R) check(smgpu_set_params(K.h, &prm), "smgpu_set_params");
    if (opt.parallel) {
        buildHalo(R);
        for (Rank& K : R) {
            HIPCHK(hipSetDevice(K.device));
            const size_t ns = (size_t)std::max(K.nSend, 1);
            HIPCHK(hipMalloc((void**)&K.sendA, ns * SMGPU_HALO_A_DOUBLES * 8));
            HIPCHK(hipMalloc((void**)&K.recvA, ns * SMGPU_HALO_A_DOUBLES * 8));
            HIPCHK(hipMalloc((void**)&K.sendF, ns * 4));
            HIPCHK(hipMalloc((void**)&K.recvF, ns * 4));
            HIPCHK(hipMalloc((void**)&K.localStats, 16));
            HIPCHK(hipMalloc((void**)&K.sendL, ns * SMGPU_HALO_L_DOUBLES * 8));
            HIPCHK(hipMalloc((void**)&K.recvL, ns * SMGPU_HALO_L_DOUBLES * 8));
            smgpu_halo_desc hd{};
            hd.nShared = (int32_t)K.sharedLocal.size(); hd.sharedLocal = K.sharedLocal.data();
            hd.nSend = K.nSend; hd.sendShared = K.sendShared.data(); hd.nRecv = K.nSend;
            hd.combOffsets = K.combOff.data(); hd.combSlots = K.combSlots.data();
            hd.sendA = K.sendA; hd.recvA = K.recvA; hd.sendF = K.sendF; hd.recvF = K.recvF; hd.localStats = K.localStats;
            hd.sendL = K.sendL; hd.recvL = K.recvL;
            check(smgpu_halo_configure(K.h, &hd), "smgpu_halo_configure");
        }
    }

    // -parallel: the reference's syncPointList calls of the set-ups are done here over the shared points (all sub-domains live
    // in this process); sharers of a global point in ascending rank order
    std::map<int64_t, std::vector<std::pair<int, int>>> sharers;   // global id -> (rank, index in the rank's shared list)
    if (opt.parallel)
        for (int r = 0; r < nRanks; ++r)
            for (size_t i = 0; i < R[r].sharedGlobal.size(); ++i) sharers[R[r].sharedGlobal[i]].push_back({r, (int)i});
    typedef int (*SharedFn)(smgpu_handle*, int32_t, int32_t, double*);
    auto syncShared = [&](SharedFn fn, const char* what, int field, int width, int op) {   // op 0 max, 1 sum (ascending rank), 2 larger magnitude folded onto own
        std::vector<std::vector<double>> v(R.size());
        for (int r = 0; r < nRanks; ++r) {
            v[r].assign(std::max<size_t>(R[r].sharedGlobal.size(), 1) * width, 0.0);
            if (!R[r].sharedGlobal.empty()) check(fn(R[r].h, field, 0, v[r].data()), what);
        }
        const std::vector<std::vector<double>> sent(v);
        for (const auto& kv : sharers) {
            const auto& sh = kv.second;
            for (const auto& me : sh) {
                double* x = &v[me.first][(size_t)me.second * width];
                if (op == 1) for (int c = 0; c < width; ++c) x[c] = 0.0;
                for (const auto& ot : sh) {
                    const double* y = &sent[ot.first][(size_t)ot.second * width];
                    if (op == 1) { for (int c = 0; c < width; ++c) x[c] = x[c] + y[c]; continue; }
                    if (ot.first == me.first) continue;
                    if (op == 0) { if (y[0] > x[0]) x[0] = y[0]; }
                    else {
                        const double mx = x[0] * x[0] + x[1] * x[1] + x[2] * x[2], my = y[0] * y[0] + y[1] * y[1] + y[2] * y[2];
                        if (!(mx >= my)) { x[0] = y[0]; x[1] = y[1]; x[2] = y[2]; }
                    }
                }
            }
        }
        for (int r = 0; r < nRanks; ++r)
            if (!R[r].sharedGlobal.empty()) check(fn(R[r].h, field, 1, v[r].data()), what);
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
            for (int r = 0; r < nRanks; ++r) check(smgpu_layers_begin(R[r].h, &ld[r], &on, &maxIter), "smgpu_layers_begin");
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
            std::printf("Target surfaces file \"%s\" stats:\nTriangles    : %zu\nVertices     : %zu\n\n", targetSurfacesFile.c_str(), surfTris.size() / 3, surfPts.size() / 3);
            if (fileExists(cd + "/" + initEdgesFile)) {
                readObjEdges(cd + "/" + initEdgesFile, initPts, initE);
                std::printf("Initial feature edges file \"%s\" stats:\n  points : %zu\n  edges  : %zu\n\n", initEdgesFile.c_str(), initPts.size() / 3, initE.size() / 2);
            }
            if (fileExists(cd + "/" + targetEdgesFile)) {
                readObjEdges(cd + "/" + targetEdgesFile, tgtPts, tgtE);
                std::printf("Target feature edges file \"%s\" stats:\n  points : %zu\n  edges  : %zu\n", targetEdgesFile.c_str(), tgtPts.size() / 3, tgtE.size() / 2);
            } else
                std::printf("WARNING: Initial feature edges will be used also as target edges, because\ndid not find file %s.\n\n", targetEdgesFile.c_str());
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
        std::printf("Distance tolerance = %g\n\n", bd[0].distanceTolerance);
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
            const double perimeter = bb[1] - bb[0] + bb[3] - bb[2] + bb[5] + bb[4];
            for (size_t r = 0; r < R.size(); ++r) {
                smgpu_boundary_info bi{};
                check(smgpu_boundary_begin(R[r].h, &bd[r], mn, perimeter, &bi), "smgpu_boundary_begin");
                if (r == 0) tot = bi;
                else {   // returnReduce sumOp, BPS.C:423-427
                    tot.enabled = tot.enabled && bi.enabled;
                    tot.nCornerPoints += bi.nCornerPoints; tot.nFeatureEdgePoints += bi.nFeatureEdgePoints;
                    tot.nSmoothingSurfacePoints += bi.nSmoothingSurfacePoints; tot.nFrozenSurfacePoints += bi.nFrozenSurfacePoints;
                }
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
        std::printf("Detected number of target edge mesh strings: %d\n\n", tot.nTargetEdgeStrings);
        std::printf("Boundary point classification summary:\n- Detected number of corner points: %d\n- Detected number of feature edge points: %d\n"
                    "- Detected number of smoothing surface points: %d\n- Detected number of frozen surface points: %d\n\n",
                    tot.nCornerPoints, tot.nFeatureEdgePoints, tot.nSmoothingSurfacePoints, tot.nFrozenSurfacePoints);
    }

    auto syncAll = [&] { for (Rank& K : R) { HIPCHK(hipSetDevice(K.device)); HIPCHK(hipDeviceSynchronize()); } };
    int32_t lDoubles = SMGPU_HALO_L_LAYERS;   // doubles per slot of the L records in use (all engines agree)
    if (opt.parallel) check(smgpu_halo_l_doubles(R[0].h, &lDoubles), "smgpu_halo_l_doubles");
    auto exchangeL = [&] {
        for (int a = 0; a < nRanks; ++a)
            for (int b = 0; b < nRanks; ++b) {
                const int c = R[a].peerCount[b];
                if (!c) continue;
                HIPCHK(hipMemcpyPeer(R[b].recvL + (size_t)R[b].peerSendBase[a] * lDoubles, R[b].device,
                                     R[a].sendL + (size_t)R[a].peerSendBase[b] * lDoubles, R[a].device, (size_t)c * lDoubles * 8));
            }
    };
    auto exchange = [&](bool isA) {
        syncAll();
        if (isA && (doLayerTreatment || doBoundarySmoothing)) exchangeL();
        for (int a = 0; a < nRanks; ++a)
            for (int b = 0; b < nRanks; ++b) {
                const int c = R[a].peerCount[b];
                if (!c) continue;
                // slots a->b on a's send side start at peerSendBase[b]; on b's recv side at b.peerSendBase[a]
                if (isA)
                    HIPCHK(hipMemcpyPeer(R[b].recvA + (size_t)R[b].peerSendBase[a] * SMGPU_HALO_A_DOUBLES, R[b].device,
                                         R[a].sendA + (size_t)R[a].peerSendBase[b] * SMGPU_HALO_A_DOUBLES, R[a].device,
                                         (size_t)c * SMGPU_HALO_A_DOUBLES * 8));
                else
                    HIPCHK(hipMemcpyPeer(R[b].recvF + R[b].peerSendBase[a], R[b].device, R[a].sendF + R[a].peerSendBase[b], R[a].device, (size_t)c * 4));
            }
        syncAll();   // device-to-device copies may return before they complete; the engines' streams are non-blocking
    };

    auto writeMesh = [&](double timeValue) {
        const std::string tn = timeName(timeValue);
        std::printf("Writing new mesh to time %s\n\n", tn.c_str());
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
    while (i < centroidalIters && !stopIteration) {
        // run up to the next write point in one engine call (no host synchronisation inside)
        long chunk = std::min(centroidalIters - i, writeInterval - (i % writeInterval));
        int32_t done = 0;
        stats.assign((size_t)chunk, smgpu_iter_stats{});
        if (!opt.parallel) {
            check(smgpu_iterate(R[0].h, (int32_t)chunk, relTol, stats.data(), &done), "smgpu_iterate");
        } else {
            for (long k = 0; k < chunk; ++k) {
                for (Rank& K : R) check(smgpu_iter_begin(K.h), "smgpu_iter_begin");
                for (Rank& K : R) check(smgpu_iter_interior(K.h), "smgpu_iter_interior");
                exchange(true);
                for (Rank& K : R) check(smgpu_iter_mid(K.h), "smgpu_iter_mid");
                for (Rank& K : R) check(smgpu_iter_ahead(K.h), "smgpu_iter_ahead");
                exchange(false);
                for (Rank& K : R) check(smgpu_iter_end(K.h), "smgpu_iter_end");
                syncAll();
                double res = 0.0, nf = 0.0;
                for (Rank& K : R) {
                    double ls[2];
                    HIPCHK(hipMemcpy(ls, K.localStats, 16, hipMemcpyDeviceToHost));
                    res = std::max(res, ls[0]);   // returnReduce maxOp, SM.C:1567
                    nf += ls[1];                  // returnReduce sumOp, SM.C:2396
                }
                stats[(size_t)k].residual = res;
                stats[(size_t)k].nFrozenPoints = (int32_t)nf;
                ++done;
                if (res < relTol) break;
            }
        }
        for (int32_t k = 0; k < done; ++k)
            std::printf("Smoothing iteration=%ld nFrozenPoints=%d residual=%g\n", i + k + 1, stats[(size_t)k].nFrozenPoints, stats[(size_t)k].residual);
        i += done;
        timeValue += done * deltaT;   // runTime++ per iteration, SM.C:2414
        const bool hitTol = done > 0 && stats[(size_t)done - 1].residual < relTol;
        if (hitTol) { std::puts("Residual reached relTol, stopping."); stopIteration = true; }
        if (i == centroidalIters) { std::puts("Maximum centroidalIters reached, stopping."); stopIteration = true; }
        // SM.C:2416: write at stop or every writeInterval iterations (not after the very first one)
        if (stopIteration || ((i % writeInterval) == 0 && i > 1)) writeMesh(timeValue);
        if (done == 0) break;
    }

    for (Rank& K : R) smgpu_destroy(K.h);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("ClockTime = %d s.\n\nEnd\n", (int)secs);
    return 0;
}
