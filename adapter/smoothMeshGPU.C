/*---------------------------------------------------------------------------*\
  smoothMeshGPU -- OpenFOAM-linked host of the MI355X smoothing engine (libsmgpu)

  COMPILE-UNTESTED IN THIS REPOSITORY: the build image has no OpenFOAM (no wmake, no
  headers).  It is the reference's utility with its smoothing loop
  (src/smoothMesh.C:2257-2437) and the set-up of the two optional features replaced by
  calls into include/smgpu.h; argList / Time / fvMesh keep doing case and mesh I/O
  (SM.C:1786-1818 createTime / createMesh, SM.C:2416-2431 write).

  Portability: only the OpenFOAM API that the reference itself uses on both lines it
  builds against (Allwmake:36-47: OpenFOAM.com v2312-v2506 and OpenFOAM.org 12) --
  argList::optionFound / optionLookupOrDefault / optionLookup / optionRead / setOption
  (SM.C:1454-1458, 1788-1918), forAll, findIndex, returnReduce, syncTools::syncPointList
  (SM.C:134) -- plus Pstream::gatherList / scatterList / scatter and PstreamBuffers.
  adapter/Allwmake detects the line exactly as the reference's Allwmake does and passes
  -DOPENFOAM_COM / -DOPENFOAM_ORG; the engine gets the matching face / cell geometry
  formulas through smgpu_set_foam_variant (the two lines differ in
  primitiveMeshFaceCentresAndAreas.C / primitiveMeshCellCentresAndVols.C).
  tests/test_adapter.py checks every smgpu_* call and struct field used here against
  include/smgpu.h and rejects argList members that exist on one line only.

  Covered: every option of the reference (SM.C:1642-1784, defaults SM.C:1857-1918),
  serial and -parallel (one rank per GPU): the loop, -layerPatches (boundary layer
  treatment) and constant/geometry/{targetSurfaces,initEdges,targetEdges}.obj with
  -smoothingPatches (boundary point smoothing, SM.C:2080-2253), the isCornerPoint /
  isFeatureEdgePoint lists of a previous run (SM.C:2039-2077).
  Shared-point records under -parallel: with -DSMGPU_WITH_RCCL (Make/options: link
  -lrccl) they move as ONE group of ncclSend / ncclRecv pairs per exchange on the
  engine's stream, device pointer to device pointer over xGMI, no host copy and no
  host synchronisation; without it they are staged through Pstream (two stream
  synchronisations per exchange -- a debugging transport).
\*---------------------------------------------------------------------------*/

#include "argList.H"
#include "Time.H"
#include "fvMesh.H"
#include "syncTools.H"
#include "processorPolyPatch.H"
#include "processorCyclicPolyPatch.H"
#include "emptyPolyPatch.H"
#include "labelIOList.H"
#include "wordReList.H"
#include "ListOps.H"
#include "triSurface.H"
#include "edgeMesh.H"
#include "PstreamBuffers.H"
#include "PstreamReduceOps.H"
#include "UOPstream.H"
#include "UIPstream.H"

#include <hip/hip_runtime.h>
#ifdef SMGPU_WITH_RCCL
#include <rccl/rccl.h>
#endif

#include <algorithm>
#include <cstring>
#include <functional>
#include <fstream>
#include <map>
#include <set>
#include <vector>

#include "smgpu.h"

using namespace Foam;

// smoothMeshCommon.H:20 REL_TOL (distanceTolerance, SM.C:1921)
static const double REL_TOL_ADAPTER = 1e-4;

namespace
{

void check(int rc)
{
    if (rc)
    {
        FatalErrorInFunction << "smgpu: " << smgpu_last_error() << exit(FatalError);
    }
}

void checkHip(hipError_t e)
{
    if (e != hipSuccess)
    {
        FatalErrorInFunction << "HIP: " << hipGetErrorString(e) << exit(FatalError);
    }
}

#ifdef SMGPU_WITH_RCCL
void checkNccl(ncclResult_t r)
{
    if (r != ncclSuccess)
    {
        FatalErrorInFunction << "RCCL: " << ncclGetErrorString(r) << exit(FatalError);
    }
}
#endif

bool fileExists(const string& name)   // SM.C uses std::ifstream for this too
{
    std::ifstream f(name.c_str());
    return f.good();
}

// isInternalPoint as SM.C:40-91: a point is internal unless a face of a
// non-processor patch uses it; empty patches are refused.
List<unsigned char> internalPointMask(const fvMesh& mesh)
{
    List<unsigned char> mask(mesh.nPoints(), static_cast<unsigned char>(1));
    forAll(mesh.boundaryMesh(), patchI)
    {
        const polyPatch& pp = mesh.boundaryMesh()[patchI];
        if (isA<processorPolyPatch>(pp)) continue;
        if (isA<emptyPolyPatch>(pp))
        {
            FatalErrorInFunction
                << "Smoothing of non-3D meshes (meshes with type empty patches) is not supported"
                << exit(FatalError);
        }
        const labelList& mp = pp.meshPoints();
        forAll(mp, i) mask[mp[i]] = 0;
    }
    return mask;
}

// SM.C:1442-1470 getPatchIdsForOption, with the reference's own option calls
labelList getPatchIdsForOption(const fvMesh& mesh, const argList& args, const word optionName)
{
    labelList patchIds;
    const polyBoundaryMesh& patches = mesh.boundaryMesh();
    if (args.optionFound(optionName))
    {
        labelHashSet patchesHashSet(patches.patchSet(wordReList(args.optionLookup(optionName)())));
        forAllConstIter(labelHashSet, patchesHashSet, iter)
        {
            const label id = iter.key();
            if (findIndex(patchIds, id) == -1) patchIds.append(id);
        }
    }
    return patchIds;
}

// patch table of smgpu_layer_desc / smgpu_boundary_desc
struct PatchTable
{
    labelList start, size;
    List<unsigned char> kind, selected;
};

PatchTable patchTable(const fvMesh& mesh, const labelList& selectedIds)
{
    PatchTable t;
    const label n = mesh.boundaryMesh().size();
    t.start.setSize(n); t.size.setSize(n); t.kind.setSize(n); t.selected.setSize(n);
    forAll(mesh.boundaryMesh(), patchI)
    {
        const polyPatch& pp = mesh.boundaryMesh()[patchI];
        t.start[patchI] = pp.start();
        t.size[patchI] = pp.size();
        t.kind[patchI] = isA<processorPolyPatch>(pp) ? 1 : isA<emptyPolyPatch>(pp) ? 2 : 0;
        t.selected[patchI] = (findIndex(selectedIds, patchI) >= 0) ? 1 : 0;
    }
    return t;
}

// Shared-point tables of this rank (what smoothmesh_amd/halo.py:HaloTables builds):
// Who shares a point with whom follows OpenFOAM's own globalPoints (what syncTools::syncPointList uses): the copies of a point
// on the two sides of a processor patch are the same point, and so is everything connected through such pairs (a rank has
// one local point per mesh point, which joins all its patches) -- and nothing else: the two sides of a baffle on different ranks
// are different shared points, or none.  The copies are matched through the patches themselves -- face i of my patch to b is
// face i of b's patch to me reversed about its first vertex (face::reverseFace), so my vertex k is its vertex (n - k) % n --
// and no pointProcAddressing is needed (a mesh made in parallel has none).  A shared point is known by the lowest (rank, local
// point) of its copies.  Slots: per peer in ascending rank, ascending key.
struct HaloTables
{
    std::vector<int32_t> sharedLocal, sendShared, combOffsets, combSlots;
    std::vector<int> peers;
    std::vector<int32_t> counts;     // per peer
    std::vector<int32_t> base;       // per peer: first send (= recv) slot
    int32_t nSend = 0, nRecv = 0;
};

HaloTables buildHalo(const fvMesh& mesh)
{
    typedef std::pair<label, label> Node;            // (rank, local point)
    HaloTables t;
    const label me = Pstream::myProcNo();
    const label nProcs = Pstream::nProcs();
    // my processor patches, flattened: {neighbour, nFaces, {n, mesh points ...} per face}*  (+ the listed vertices' coordinates)
    std::vector<label> flat;
    std::vector<point> flatPts;
    forAll(mesh.boundaryMesh(), patchI)
    {
        const polyPatch& pp = mesh.boundaryMesh()[patchI];
        if (!isA<processorPolyPatch>(pp)) continue;
        // (isA<processorPolyPatch> is also true for a processorCyclicPolyPatch: the copies of a point across a cyclic are NOT one
        // point -- their positions differ by the patch transform -- and nothing here transforms positions or sums; the C++
        // front-end refuses such cases too)
        if (isA<processorCyclicPolyPatch>(pp))
            FatalErrorInFunction << "patch " << pp.name() << " is a processorCyclic patch: cyclic coupling between sub-domains is not supported"
                                 << exit(FatalError);
        flat.push_back(refCast<const processorPolyPatch>(pp).neighbProcNo());
        flat.push_back(pp.size());
        for (label i = 0; i < pp.size(); ++i)
        {
            const face& f = mesh.faces()[pp.start() + i];
            flat.push_back(f.size());
            forAll(f, k) { flat.push_back(f[k]); flatPts.push_back(mesh.points()[f[k]]); }
        }
    }
    List<labelList> all(nProcs);
    all[me].setSize(static_cast<label>(flat.size()));
    forAll(all[me], i) all[me][i] = flat[static_cast<size_t>(i)];
    Pstream::gatherList(all);
    Pstream::scatterList(all);
    List<pointField> allPts(nProcs);
    allPts[me].setSize(static_cast<label>(flatPts.size()));
    forAll(allPts[me], i) allPts[me][i] = flatPts[static_cast<size_t>(i)];
    Pstream::gatherList(allPts);
    Pstream::scatterList(allPts);

    // per rank and neighbour: where each face's {n, points...} record starts (and where its coordinates start)
    std::vector<std::map<label, std::vector<label>>> faceAt(static_cast<size_t>(nProcs)), coordAt(static_cast<size_t>(nProcs));
    for (label o = 0; o < nProcs; ++o)
    {
        const labelList& v = all[o];
        label co = 0;
        for (label k = 0; k + 1 < v.size();)
        {
            const label nb = v[k], nF = v[k + 1];
            k += 2;
            std::vector<label>& at = faceAt[static_cast<size_t>(o)][nb];
            std::vector<label>& cat = coordAt[static_cast<size_t>(o)][nb];
            for (label f = 0; f < nF; ++f) { at.push_back(k); cat.push_back(co); co += v[k]; k += 1 + v[k]; }
        }
    }
    std::map<Node, Node> parent;                     // the root of a component is its lowest node
    std::function<Node(Node)> find = [&](Node a)
    {
        std::map<Node, Node>::iterator it = parent.find(a);
        if (it == parent.end()) { parent[a] = a; return a; }
        if (it->second == a) return a;
        const Node root = find(it->second);
        parent[a] = root;
        return root;
    };
    for (label a = 0; a < nProcs; ++a)
    {
        for (const auto& kv : faceAt[static_cast<size_t>(a)])
        {
            const label b = kv.first;
            if (b <= a || b >= nProcs) continue;
            const auto it = faceAt[static_cast<size_t>(b)].find(a);
            if (it == faceAt[static_cast<size_t>(b)].end() || it->second.size() != kv.second.size())
                FatalErrorInFunction << "processor patches " << a << " <-> " << b << " do not match" << exit(FatalError);
            for (size_t f = 0; f < kv.second.size(); ++f)
            {
                const label pa = kv.second[f], pb = it->second[f];
                const label nv = all[a][pa];
                if (all[b][pb] != nv)
                    FatalErrorInFunction << "processor patches " << a << " <-> " << b << ": face sizes differ" << exit(FatalError);
                // vertex k of one side is vertex (n - k) mod n of the other (the face reversed about its first vertex): nothing but the
                // coordinates can tell whether the two sides really list their faces that way -- OpenFOAM's matchTolerance
                const label ca = coordAt[static_cast<size_t>(a)][b][f], cb = coordAt[static_cast<size_t>(b)][a][f];
                scalar ext = 0;
                for (label k = 1; k < nv; ++k) ext = max(ext, mag(allPts[a][ca + k] - allPts[a][ca]));
                for (label k = 0; k < nv; ++k)
                {
                    if (mag(allPts[a][ca + k] - allPts[b][cb + (nv - k) % nv]) > 1e-4 * ext)
                        FatalErrorInFunction << "processor patches " << a << " <-> " << b << ": face " << label(f) << " vertex " << k
                                             << " does not coincide with its copy on the other side" << exit(FatalError);
                    const Node x = find(Node(a, all[a][pa + 1 + k])), y = find(Node(b, all[b][pb + 1 + (nv - k) % nv]));
                    if (x != y) { if (x < y) parent[y] = x; else parent[x] = y; }
                }
            }
        }
    }
    // the components this rank takes part in: key = root, my copy, the other ranks
    std::map<Node, label> localOf;
    std::map<Node, std::vector<int>> sharers;        // key -> the OTHER ranks of the group (ascending)
    std::vector<std::vector<Node>> with(static_cast<size_t>(nProcs));
    {
        std::vector<Node> nodes;
        for (const auto& kv : parent) nodes.push_back(kv.first);
        for (const Node& nd : nodes)
        {
            if (nd.first != me) continue;
            const Node key = find(nd);
            // a rank has ONE local point per mesh point: two of them in one component mean the patches do not pair up as assumed
            if (localOf.count(key) && localOf[key] != nd.second)
                FatalErrorInFunction << "local points " << localOf[key] << " and " << nd.second << " are matched to the same shared point" << exit(FatalError);
            localOf[key] = nd.second;
        }
        for (const Node& nd : nodes)                 // (ascending rank: the map is ordered by (rank, point))
        {
            if (nd.first == me) continue;
            const Node key = find(nd);
            if (!localOf.count(key)) continue;
            sharers[key].push_back(static_cast<int>(nd.first));
            with[static_cast<size_t>(nd.first)].push_back(key);
        }
        for (auto& w : with) std::sort(w.begin(), w.end());
    }
    std::map<Node, int32_t> sharedIndex;
    for (const auto& kv : sharers)
    {
        sharedIndex[kv.first] = static_cast<int32_t>(t.sharedLocal.size());
        t.sharedLocal.push_back(static_cast<int32_t>(localOf[kv.first]));
    }
    std::map<int, int32_t> base;
    for (label r = 0; r < nProcs; ++r)
    {
        const std::vector<Node>& w = with[static_cast<size_t>(r)];
        if (w.empty()) continue;
        t.peers.push_back(r);
        t.counts.push_back(static_cast<int32_t>(w.size()));
        t.base.push_back(t.nSend);
        base[r] = t.nSend;
        for (const Node& g : w) t.sendShared.push_back(sharedIndex[g]);
        t.nSend += static_cast<int32_t>(w.size());
    }
    t.nRecv = t.nSend;                     // the lists are symmetric
    t.combOffsets.push_back(0);
    for (const auto& kv : sharers)
    {
        std::vector<int> ranks(kv.second);
        ranks.push_back(me);
        std::sort(ranks.begin(), ranks.end());
        for (int r : ranks)
        {
            if (r == me) { t.combSlots.push_back(-1); continue; }
            const std::vector<Node>& w = with[static_cast<size_t>(r)];
            const int32_t k = static_cast<int32_t>(std::lower_bound(w.begin(), w.end(), kv.first) - w.begin());
            t.combSlots.push_back(base[r] + k);
        }
        t.combOffsets.push_back(static_cast<int32_t>(t.combSlots.size()));
    }
    return t;
}

// One exchange = for every rank that shares points with this one, its slots of the send
// buffers against the matching slots of the receive buffers (syncTools::syncPointList's
// transport, SM.C:134-148, 402-478, 2374; OBB.C:184-198, 490-496).
struct Part { const void* send; void* recv; size_t bytesPerSlot; };

struct Exchanger
{
    const HaloTables& t;
    hipStream_t stream;
#ifdef SMGPU_WITH_RCCL
    ncclComm_t comm = nullptr;
#endif
    std::vector<char> hSend, hRecv;

    Exchanger(const HaloTables& tables, hipStream_t s) : t(tables), stream(s)
    {
#ifdef SMGPU_WITH_RCCL
        // one communicator over all ranks; the id travels through Pstream as a list of labels
        ncclUniqueId id;
        std::memset(&id, 0, sizeof(id));
        if (Pstream::master()) checkNccl(ncclGetUniqueId(&id));
        labelList idWords((sizeof(id) + sizeof(label) - 1) / sizeof(label), 0);
        if (Pstream::master()) std::memcpy(idWords.begin(), &id, sizeof(id));
        Pstream::scatter(idWords);
        std::memcpy(&id, idWords.begin(), sizeof(id));
        checkNccl(ncclCommInitRank(&comm, Pstream::nProcs(), id, Pstream::myProcNo()));
#endif
    }

    ~Exchanger()
    {
#ifdef SMGPU_WITH_RCCL
        if (comm)
        {
            (void)hipStreamSynchronize(stream);
            (void)ncclCommDestroy(comm);
        }
#endif
    }

    void operator()(std::initializer_list<Part> parts)
    {
        if (!t.nSend) return;
#ifdef SMGPU_WITH_RCCL
        // device pointer to device pointer, one collective group on the engine's stream: nothing waits on the host
        checkNccl(ncclGroupStart());
        for (const Part& pt : parts)
        {
            for (size_t i = 0; i < t.peers.size(); ++i)
            {
                const size_t off = size_t(t.base[i]) * pt.bytesPerSlot, n = size_t(t.counts[i]) * pt.bytesPerSlot;
                checkNccl(ncclSend(static_cast<const char*>(pt.send) + off, n, ncclChar, t.peers[i], comm, stream));
                checkNccl(ncclRecv(static_cast<char*>(pt.recv) + off, n, ncclChar, t.peers[i], comm, stream));
            }
        }
        checkNccl(ncclGroupEnd());
#else
        for (const Part& pt : parts)
        {
            hSend.resize(size_t(t.nSend) * pt.bytesPerSlot);
            hRecv.resize(size_t(t.nRecv) * pt.bytesPerSlot);
            checkHip(hipMemcpyAsync(hSend.data(), pt.send, hSend.size(), hipMemcpyDeviceToHost, stream));
            checkHip(hipStreamSynchronize(stream));
            PstreamBuffers pBufs(Pstream::commsTypes::nonBlocking);
            for (size_t i = 0; i < t.peers.size(); ++i)
            {
                UOPstream to(t.peers[i], pBufs);
                to.write(hSend.data() + size_t(t.base[i]) * pt.bytesPerSlot, std::streamsize(size_t(t.counts[i]) * pt.bytesPerSlot));
            }
            pBufs.finishedSends();
            for (size_t i = 0; i < t.peers.size(); ++i)
            {
                UIPstream from(t.peers[i], pBufs);
                from.read(hRecv.data() + size_t(t.base[i]) * pt.bytesPerSlot, std::streamsize(size_t(t.counts[i]) * pt.bytesPerSlot));
            }
            checkHip(hipMemcpyAsync(pt.recv, hRecv.data(), hRecv.size(), hipMemcpyHostToDevice, stream));
            checkHip(hipStreamSynchronize(stream));   // hRecv is reused by the next part
        }
#endif
    }
};

// The reference's syncTools::syncPointList calls of the two set-ups (OBB.C:124-130, 184-198, 359-365): the engine hands
// out its values at the shared points (order of HaloTables::sharedLocal), OpenFOAM itself synchronises them on a full
// point list exactly as the reference does, the combined values go back.
typedef int (*SharedFn)(smgpu_handle*, int32_t, int32_t, double*);

template<class Type, class CombineOp>
void syncShared
(
    const fvMesh& mesh, smgpu_handle* h, const HaloTables& t, SharedFn fn, int32_t field, int width,
    const CombineOp& cop, const Type& nullValue,
    Type (*pack)(const double*), void (*unpack)(const Type&, double*)
)
{
    std::vector<double> buf(std::max<size_t>(t.sharedLocal.size() * size_t(width), 1));
    check(fn(h, field, 0, buf.data()));
    List<Type> full(mesh.nPoints(), nullValue);
    for (size_t i = 0; i < t.sharedLocal.size(); ++i) full[t.sharedLocal[i]] = pack(&buf[i * size_t(width)]);
    syncTools::syncPointList(mesh, full, cop, nullValue);
    for (size_t i = 0; i < t.sharedLocal.size(); ++i) unpack(full[t.sharedLocal[i]], &buf[i * size_t(width)]);
    check(fn(h, field, 1, buf.data()));
}

scalar packScalar(const double* v) { return v[0]; }
void unpackScalar(const scalar& s, double* v) { v[0] = s; }
vector packVector(const double* v) { return vector(v[0], v[1], v[2]); }
void unpackVector(const vector& s, double* v) { v[0] = s.x(); v[1] = s.y(); v[2] = s.z(); }

}  // namespace


int main(int argc, char *argv[])
{
    argList::addNote("Move internal mesh points to increase mesh quality on an AMD MI355X (libsmgpu); options as smoothMesh");
    // SM.C:1642-1784
    argList::addOption("time", "time", "Specify the time (default is latest)");
    argList::addBoolOption("overwrite", "Overwrite the mesh of the start time");
    argList::addOption("centroidalIters", "label", "Maximum number of centroidal smoothing iterations (default 1000)");
    argList::addOption("maxStepLength", "double", "Maximum absolute step length applied in smoothing (default 0.3 * minEdgeLength)");
    argList::addOption("relStepFrac", "double", "Relative step fraction of the centroidal step (default 0.5)");
    argList::addOption("minEdgeLength", "double", "Edges shorter than this are not shortened further (default 0.5 * shortest edge)");
    argList::addOption("totalMinFreeze", "bool", "Freeze all points of edges shorter than minEdgeLength (default false)");
    argList::addOption("edgeAngleConstraint", "bool", "Prohibit the decrease of small edge-edge angles (default true)");
    argList::addOption("faceAngleConstraint", "bool", "Prohibit the deterioration of face-face angles (default true)");
    argList::addOption("minAngle", "double", "Angle (degrees) below which angles may not decrease (default 35)");
    argList::addOption("maxAngle", "double", "Angle (degrees) above which face angles may not increase (default 160)");
    argList::addOption("layerPatches", "wordRes", "Patches with boundary layer treatment (default none)");
    argList::addOption("layerMaxBlendingFraction", "double", "default 0.3");
    argList::addOption("layerEdgeLength", "double", "default minEdgeLength");
    argList::addOption("layerExpansionRatio", "double", "default 1.3");
    argList::addOption("minLayers", "label", "default 1");
    argList::addOption("maxLayers", "label", "default 4");
    argList::addOption("smoothingPatches", "wordRes", "Patches whose points are smoothed along the boundary (default all)");
    argList::addOption("internalSmoothingBlendingFraction", "double", "default 0");
    argList::addOption("relTol", "double", "Relative tolerance for stopping the smoothing iterations (default: 0.02)");
    argList::addOption("writeInterval", "label", "Interval to write mesh during iterations (default value: Same as centroidalIters)");
    argList::addOption("device", "label", "HIP device ordinal (default: rank modulo the number of devices)");

    #include "setRootCase.H"
    #include "createTime.H"

    const bool overwrite = args.optionFound("overwrite");

    // SM.C:1791-1803
    if (args.optionFound("time"))
    {
        if (args["time"] == "constant")
        {
            runTime.setTime(instant(0, "constant"), 0);
        }
        else
        {
            const scalar timeValue = args.optionRead<scalar>("time");
            runTime.setTime(instant(timeValue), 0);
        }
    }
    if (runTime.deltaTValue() < VSMALL)
    {
        FatalError << "Time step (deltaT) value " << runTime.deltaTValue()
                   << " specified in controlDict is too small" << endl << abort(FatalError);
    }

    // SM.C:1814-1818
    #ifdef OPENFOAM_ORG
        #include "createMesh.H"
    #else
        #include "createMeshNoClear.H"
    #endif

    const word oldInstance = mesh.pointsInstance();

    if (sizeof(label) != sizeof(int32_t) || sizeof(scalar) != sizeof(double))
    {
        FatalErrorInFunction << "libsmgpu needs WM_LABEL_SIZE=32 and WM_PRECISION_OPTION=DP" << exit(FatalError);
    }

    // SM.C:1820-1852
    labelList layerPatchIds = getPatchIdsForOption(mesh, args, "layerPatches");
    if (! args.optionFound("smoothingPatches"))
    {
        args.setOption("smoothingPatches", "(\".*\")");
    }
    labelList smoothingPatchIds = getPatchIdsForOption(mesh, args, "smoothingPatches");

    // ---- hand the mesh over (include/smgpu.h, smgpu_mesh_desc) ----------------------------
    const faceList& faces = mesh.faces();
    labelList faceOffsets(faces.size() + 1, 0);
    forAll(faces, fI) faceOffsets[fI + 1] = faceOffsets[fI] + faces[fI].size();
    labelList facePoints(faceOffsets[faces.size()]);
    forAll(faces, fI) forAll(faces[fI], k) facePoints[faceOffsets[fI] + k] = faces[fI][k];
    const List<unsigned char> isInternalPoint(internalPointMask(mesh));

    int nDev = 0;
    checkHip(hipGetDeviceCount(&nDev));
    const label device = args.optionLookupOrDefault("device", label(Pstream::parRun() ? Pstream::myProcNo() % max(nDev, 1) : 0));

    smgpu_mesh_desc d = {};      // (every field zero: fields added to the struct later stay defined)
    d.nPoints = mesh.nPoints(); d.nCells = mesh.nCells(); d.nFaces = mesh.nFaces(); d.nInternalFaces = mesh.nInternalFaces();
    d.points = reinterpret_cast<const double*>(mesh.points().begin());      // Vector<double>: 3 packed doubles
    d.faceOffsets = faceOffsets.begin(); d.facePoints = facePoints.begin();
    d.owner = mesh.faceOwner().begin(); d.neighbour = mesh.faceNeighbour().begin();
    d.isInternalPoint = isInternalPoint.begin();
    d.isSmoothingSurfacePoint = nullptr;       // decided by the boundary smoothing set-up below (BPS.C:404-420)
    d.device = device; d.stream = nullptr; d.useCallerStream = 0;
    smgpu_handle* h = nullptr;
    check(smgpu_create(&d, &h));
    // the face / cell geometry of the OpenFOAM line this host is compiled against (Allwmake passes the define)
    #if defined(OPENFOAM_ORG)
    check(smgpu_set_foam_variant(h, SMGPU_FOAM_ORG));
    #elif defined(OPENFOAM_COM) || defined(OPENFOAM)
    check(smgpu_set_foam_variant(h, SMGPU_FOAM_COM));
    #else
    #error "compile through adapter/Allwmake (or pass -DOPENFOAM_COM / -DOPENFOAM_ORG): the two OpenFOAM lines compute face centres differently"
    #endif

    // ---- parameters, defaults as SM.C:1854-1918 ---------------------------------------------
    double meshMinEdgeLength = 0, meshMaxEdgeLength = 0;
    check(smgpu_mesh_stats(h, &meshMinEdgeLength, &meshMaxEdgeLength));
    reduce(meshMinEdgeLength, minOp<scalar>());                             // SM.C:1527
    reduce(meshMaxEdgeLength, maxOp<scalar>());
    double minEdgeLength = args.optionLookupOrDefault("minEdgeLength", 0.5 * meshMinEdgeLength);
    double maxStepLength = args.optionLookupOrDefault("maxStepLength", 0.3 * minEdgeLength);
    if (maxStepLength > 0.5 * minEdgeLength)
    {
        Info << "WARNING: The maximum allowed step length is more "
             << "than half of the minimum edge length! This may "
             << "cause unstability in smoothing." << endl << endl;
    }
    smgpu_params prm = {};      // (every field zero: fields added to the struct later stay defined)
    prm.maxStepLength = maxStepLength;
    prm.relStepFrac = args.optionLookupOrDefault("relStepFrac", 0.5);
    prm.minEdgeLength = minEdgeLength;
    prm.totalMinFreeze = args.optionLookupOrDefault("totalMinFreeze", false);
    prm.minAngle = args.optionLookupOrDefault("minAngle", 35.0);
    prm.maxAngle = args.optionLookupOrDefault("maxAngle", 160.0);
    prm.edgeAngleConstraint = args.optionLookupOrDefault("edgeAngleConstraint", true);
    prm.faceAngleConstraint = args.optionLookupOrDefault("faceAngleConstraint", true);
    check(smgpu_set_params(h, &prm));
    const double layerMaxBlendingFraction = args.optionLookupOrDefault("layerMaxBlendingFraction", 0.3);
    const double layerEdgeLength = args.optionLookupOrDefault("layerEdgeLength", minEdgeLength);
    const double layerExpansionRatio = args.optionLookupOrDefault("layerExpansionRatio", 1.3);
    const label minLayers = args.optionLookupOrDefault("minLayers", 1);
    const label maxLayers = args.optionLookupOrDefault("maxLayers", 4);
    const double internalSmoothingBlendingFraction = args.optionLookupOrDefault("internalSmoothingBlendingFraction", 0.0);
    const double relTol = args.optionLookupOrDefault("relTol", 0.02);
    const label centroidalIters = args.optionLookupOrDefault("centroidalIters", 1000);
    const label writeInterval = args.optionLookupOrDefault("writeInterval", centroidalIters);      // SM.C:1918
    const double distanceTolerance = REL_TOL_ADAPTER * min(meshMinEdgeLength, layerEdgeLength);     // SM.C:1921

    const string initEdgesFileString("constant/geometry/initEdges.obj");                             // SM.C:1924-1930
    const string targetEdgesFileString("constant/geometry/targetEdges.obj");
    const string targetSurfacesFileString("constant/geometry/targetSurfaces.obj");

    // ---- -parallel: shared-point tables, exchange buffers, halo (before the set-ups that synchronise) ---------
    HaloTables halo;
    void *sendA = nullptr, *recvA = nullptr, *sendF = nullptr, *recvF = nullptr, *sendL = nullptr, *recvL = nullptr, *localStats = nullptr;
    hipStream_t stream = nullptr;
    {
        void* vs = nullptr;
        check(smgpu_get_stream(h, &vs));
        stream = static_cast<hipStream_t>(vs);
    }
    if (Pstream::parRun())
    {
        halo = buildHalo(mesh);
        checkHip(hipSetDevice(device));
        const size_t nS = size_t(max(halo.nSend, 1)), nR = size_t(max(halo.nRecv, 1));
        checkHip(hipMalloc(&sendA, nS * SMGPU_HALO_A_DOUBLES * sizeof(double)));
        checkHip(hipMalloc(&recvA, nR * SMGPU_HALO_A_DOUBLES * sizeof(double)));
        checkHip(hipMalloc(&sendL, nS * SMGPU_HALO_L_DOUBLES * sizeof(double)));
        checkHip(hipMalloc(&recvL, nR * SMGPU_HALO_L_DOUBLES * sizeof(double)));
        checkHip(hipMalloc(&sendF, nS * sizeof(int32_t)));
        checkHip(hipMalloc(&recvF, nR * sizeof(int32_t)));
        checkHip(hipMalloc(&localStats, 2 * sizeof(double)));
        smgpu_halo_desc hd = {};      // (every field zero: fields added to the struct later stay defined)
        hd.nShared = int32_t(halo.sharedLocal.size()); hd.sharedLocal = halo.sharedLocal.data();
        hd.nSend = halo.nSend; hd.sendShared = halo.sendShared.data();
        hd.nRecv = halo.nRecv; hd.combOffsets = halo.combOffsets.data(); hd.combSlots = halo.combSlots.data();
        hd.sendA = sendA; hd.recvA = recvA; hd.sendF = sendF; hd.recvF = recvF; hd.localStats = localStats;
        hd.sendL = sendL; hd.recvL = recvL;
        hd.useExchangeStream = 0; hd.exchangeStream = nullptr;      // exchanges in order on the engine's stream
        check(smgpu_halo_configure(h, &hd));
    }

    // ---- boundary layer treatment, SM.C:2024-2035, 2186-2221 ----------------------------------
    bool doLayerTreatment = false;
    const PatchTable layerTable(patchTable(mesh, layerPatchIds));
    smgpu_layer_desc ld = {};      // (every field zero: fields added to the struct later stay defined)
    ld.nPatches = layerTable.start.size(); ld.patchStart = layerTable.start.begin(); ld.patchSize = layerTable.size.begin();
    ld.patchKind = layerTable.kind.begin(); ld.isLayerPatch = layerTable.selected.begin();
    ld.layerMaxBlendingFraction = layerMaxBlendingFraction;
    ld.layerEdgeLength = layerEdgeLength;
    ld.layerExpansionRatio = layerExpansionRatio;
    ld.minLayers = minLayers; ld.maxLayers = maxLayers;
    if ((layerPatchIds.size() > 0) and (layerMaxBlendingFraction > SMALL))
    {
        int32_t enabled = 0;
        if (!Pstream::parRun())
        {
            check(smgpu_set_layers(h, &ld, &enabled));
        }
        else
        {
            // the set-up in steps; between them the reference's syncPointList calls, done by OpenFOAM itself
            int32_t maxIter = 0;
            check(smgpu_layers_begin(h, &ld, &enabled, &maxIter));
            for (label it = 0; enabled && it < maxIter; ++it)
            {
                check(smgpu_layers_step(h, SMGPU_LAYERS_HOPS_SWEEP, 0));
                syncShared<scalar>(mesh, h, halo, smgpu_layers_shared, SMGPU_LAYERS_F_HOPS, 1, maxEqOp<scalar>(), scalar(-1),
                                   packScalar, unpackScalar);                                   // OBB.C:124-130
            }
            if (enabled)
            {
                check(smgpu_layers_step(h, SMGPU_LAYERS_NORMALS_ACCUMULATE, 0));
                // normal (3) + face count (1): two plusEqOp syncs in the reference, OBB.C:184-198
                {
                    std::vector<double> buf(std::max<size_t>(halo.sharedLocal.size() * 4, 1));
                    check(smgpu_layers_shared(h, SMGPU_LAYERS_F_NORMALS_COUNT, 0, buf.data()));
                    vectorField nrm(mesh.nPoints(), vector::zero);
                    scalarField cnt(mesh.nPoints(), 0.0);
                    for (size_t i = 0; i < halo.sharedLocal.size(); ++i)
                    {
                        nrm[halo.sharedLocal[i]] = vector(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2]);
                        cnt[halo.sharedLocal[i]] = buf[4 * i + 3];
                    }
                    syncTools::syncPointList(mesh, nrm, plusEqOp<vector>(), vector::zero);
                    syncTools::syncPointList(mesh, cnt, plusEqOp<scalar>(), scalar(0));
                    for (size_t i = 0; i < halo.sharedLocal.size(); ++i)
                    {
                        const vector& v = nrm[halo.sharedLocal[i]];
                        buf[4 * i] = v.x(); buf[4 * i + 1] = v.y(); buf[4 * i + 2] = v.z(); buf[4 * i + 3] = cnt[halo.sharedLocal[i]];
                    }
                    check(smgpu_layers_shared(h, SMGPU_LAYERS_F_NORMALS_COUNT, 1, buf.data()));
                }
                check(smgpu_layers_step(h, SMGPU_LAYERS_NORMALS_FINISH, 0));
                for (label it = 1; it <= maxIter; ++it)
                {
                    check(smgpu_layers_step(h, SMGPU_LAYERS_PROPAGATE_SWEEP, it));
                    syncShared<vector>(mesh, h, halo, smgpu_layers_shared, SMGPU_LAYERS_F_NORMALS, 3, maxMagSqrEqOp<vector>(),
                                       vector::zero, packVector, unpackVector);                 // OBB.C:359-365
                }
                check(smgpu_layers_step(h, SMGPU_LAYERS_FINISH, 0));
            }
        }
        doLayerTreatment = enabled != 0;
    }
    if (doLayerTreatment) Info << "Enabled boundary layer treatment" << endl << endl;
    else Info << "Boundary layer treatment is disabled. Either no layerPatches were specified or boundaryMaxBlendingFraction is zero" << endl << endl;

    // ---- boundary point smoothing, SM.C:2037-2253 -----------------------------------------------
    labelIOList isCornerPointIO
    (
        IOobject("isCornerPoint", runTime.name(), mesh, IOobject::READ_IF_PRESENT, IOobject::AUTO_WRITE, true),
        labelList(mesh.nPoints(), 0)
    );
    labelIOList isFeatureEdgePointIO
    (
        IOobject("isFeatureEdgePoint", runTime.name(), mesh, IOobject::READ_IF_PRESENT, IOobject::AUTO_WRITE, true),
        labelList(mesh.nPoints(), 0)
    );
    const bool labelIOListsHaveData = (findIndex(isCornerPointIO, 1) >= 0) or (findIndex(isFeatureEdgePointIO, 1) >= 0);

    bool doBoundarySmoothing = false;
    if ((fileExists(targetSurfacesFileString)) and
        ((fileExists(initEdgesFileString)) or (labelIOListsHaveData)) and
        (smoothingPatchIds.size() > 0))
    {
        // the contents of the three files as flat arrays (SM.C:2131-2160)
        const fileName surfFile(targetSurfacesFileString);      // (a named object: `triSurface surf(fileName(x));` declares a function)
        triSurface surf(surfFile);
        List<double> surfPts(3 * surf.points().size());
        forAll(surf.points(), i) { surfPts[3*i] = surf.points()[i].x(); surfPts[3*i+1] = surf.points()[i].y(); surfPts[3*i+2] = surf.points()[i].z(); }
        labelList surfTris(3 * surf.size());
        forAll(surf, i) { surfTris[3*i] = surf[i][0]; surfTris[3*i+1] = surf[i][1]; surfTris[3*i+2] = surf[i][2]; }
        List<double> iePts, tePts;
        labelList ieEdges, teEdges;
        if (fileExists(initEdgesFileString))
        {
            const fileName emFile(initEdgesFileString);
            edgeMesh em(emFile);
            iePts.setSize(3 * em.points().size());
            forAll(em.points(), i) { iePts[3*i] = em.points()[i].x(); iePts[3*i+1] = em.points()[i].y(); iePts[3*i+2] = em.points()[i].z(); }
            ieEdges.setSize(2 * em.edges().size());
            forAll(em.edges(), i) { ieEdges[2*i] = em.edges()[i].start(); ieEdges[2*i+1] = em.edges()[i].end(); }
        }
        if (fileExists(targetEdgesFileString))
        {
            const fileName emFile(targetEdgesFileString);
            edgeMesh em(emFile);
            tePts.setSize(3 * em.points().size());
            forAll(em.points(), i) { tePts[3*i] = em.points()[i].x(); tePts[3*i+1] = em.points()[i].y(); tePts[3*i+2] = em.points()[i].z(); }
            teEdges.setSize(2 * em.edges().size());
            forAll(em.edges(), i) { teEdges[2*i] = em.edges()[i].start(); teEdges[2*i+1] = em.edges()[i].end(); }
        }
        const PatchTable smoothTable(patchTable(mesh, smoothingPatchIds));
        smgpu_boundary_desc bd = {};      // (every field zero: fields added to the struct later stay defined)
        bd.nPatches = smoothTable.start.size(); bd.patchStart = smoothTable.start.begin(); bd.patchSize = smoothTable.size.begin();
        bd.patchKind = smoothTable.kind.begin(); bd.isSmoothingPatch = smoothTable.selected.begin();
        bd.nInitEdgePoints = iePts.size() / 3; bd.initEdgePoints = iePts.begin();
        bd.nInitEdges = ieEdges.size() / 2; bd.initEdges = ieEdges.begin();
        bd.nTargetEdgePoints = tePts.size() / 3; bd.targetEdgePoints = tePts.begin();
        bd.nTargetEdges = teEdges.size() / 2; bd.targetEdges = teEdges.begin();            // 0 = the initial edges are the targets (SM.C:2154-2160)
        bd.nSurfacePoints = surfPts.size() / 3; bd.surfacePoints = surfPts.begin();
        bd.nSurfaceTriangles = surfTris.size() / 3; bd.surfaceTriangles = surfTris.begin();
        bd.isCornerPointIO = labelIOListsHaveData ? isCornerPointIO.begin() : nullptr;      // SM.C:2066-2077
        bd.isFeatureEdgePointIO = labelIOListsHaveData ? isFeatureEdgePointIO.begin() : nullptr;
        bd.distanceTolerance = distanceTolerance;
        bd.internalSmoothingBlendingFraction = internalSmoothingBlendingFraction;
        smgpu_boundary_info bi = {};      // (every field zero: fields added to the struct later stay defined)
        if (!Pstream::parRun())
        {
            check(smgpu_set_boundary_smoothing(h, &bd, &bi));
        }
        else
        {
            // getMeshStats' reductions (SM.C:1528-1538), then the set-up in steps with its syncPointList calls
            double mn = 0, bb[6];
            check(smgpu_boundary_stats(h, &mn, bb));
            reduce(mn, minOp<scalar>());
            for (int k = 0; k < 6; k += 2) { reduce(bb[k], minOp<scalar>()); reduce(bb[k + 1], maxOp<scalar>()); }
            const double perimeter = (bb[1] - bb[0]) + (bb[3] - bb[2]) + (bb[5] + bb[4]);        // SM.C:1538 as written
            check(smgpu_boundary_begin(h, &bd, mn, perimeter, &bi));
            if (bi.enabled)
            {
                for (int sweep = 0; sweep < 2; ++sweep)
                {
                    check(smgpu_boundary_step(h, SMGPU_BOUNDARY_HOPS_SWEEP));
                    syncShared<scalar>(mesh, h, halo, smgpu_boundary_shared, SMGPU_BOUNDARY_F_HOPS, 1, maxEqOp<scalar>(), scalar(-1),
                                       packScalar, unpackScalar);                               // OBB.C:124-130
                }
                check(smgpu_boundary_step(h, SMGPU_BOUNDARY_TABLES));
                check(smgpu_boundary_step(h, SMGPU_BOUNDARY_NORMALS_ACCUMULATE));
                {
                    std::vector<double> buf(std::max<size_t>(halo.sharedLocal.size() * 4, 1));
                    check(smgpu_boundary_shared(h, SMGPU_BOUNDARY_F_NORMALS_COUNT, 0, buf.data()));
                    vectorField nrm(mesh.nPoints(), vector::zero);
                    scalarField cnt(mesh.nPoints(), 0.0);
                    for (size_t i = 0; i < halo.sharedLocal.size(); ++i)
                    {
                        nrm[halo.sharedLocal[i]] = vector(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2]);
                        cnt[halo.sharedLocal[i]] = buf[4 * i + 3];
                    }
                    syncTools::syncPointList(mesh, nrm, plusEqOp<vector>(), vector::zero);      // OBB.C:184-198
                    syncTools::syncPointList(mesh, cnt, plusEqOp<scalar>(), scalar(0));
                    for (size_t i = 0; i < halo.sharedLocal.size(); ++i)
                    {
                        const vector& v = nrm[halo.sharedLocal[i]];
                        buf[4 * i] = v.x(); buf[4 * i + 1] = v.y(); buf[4 * i + 2] = v.z(); buf[4 * i + 3] = cnt[halo.sharedLocal[i]];
                    }
                    check(smgpu_boundary_shared(h, SMGPU_BOUNDARY_F_NORMALS_COUNT, 1, buf.data()));
                }
                check(smgpu_boundary_step(h, SMGPU_BOUNDARY_NORMALS_FINISH));
            }
        }
        doBoundarySmoothing = bi.enabled != 0;
        if (doBoundarySmoothing)
        {
            Info << "Boundary point classification summary:" << nl
                 << "- Detected number of corner points: " << returnReduce(label(bi.nCornerPoints), sumOp<label>()) << nl
                 << "- Detected number of feature edge points: " << returnReduce(label(bi.nFeatureEdgePoints), sumOp<label>()) << nl
                 << "- Detected number of smoothing surface points: " << returnReduce(label(bi.nSmoothingSurfacePoints), sumOp<label>()) << nl
                 << "- Detected number of frozen surface points: " << returnReduce(label(bi.nFrozenSurfacePoints), sumOp<label>()) << nl << endl;
        }
    }
    if (doBoundarySmoothing) Info << "Enabled boundary point smoothing" << endl << endl;
    else
    {
        Info << "Boundary point smoothing is disabled. Missing smoothingPatches, or one or both of files:" << endl
             << targetSurfacesFileString << endl << initEdgesFileString << endl << endl;
    }
    if ((doLayerTreatment) and (! doBoundarySmoothing))
    {
        Info << "WARNING: Boundary layer treatment will be done without boundary point smoothing. This can result in distorted boundary cells." << endl << endl;
    }

    // SM.C:2416-2431
    auto writeMesh = [&]()
    {
        pointField newPoints(mesh.nPoints());
        check(smgpu_get_points(h, reinterpret_cast<double*>(newPoints.begin())));
        mesh.movePoints(newPoints);          // OpenFOAM's own geometry is only needed for the write
        if (doBoundarySmoothing)             // the AUTO_WRITE lists of SM.C:2039-2064
        {
            check(smgpu_get_boundary_classification(h, isCornerPointIO.begin(), isFeatureEdgePointIO.begin()));
        }
        if (overwrite)
        {
            mesh.setInstance(oldInstance);
        }
        IOstream::defaultPrecision(max(10u, IOstream::defaultPrecision()));   // SM.C:2425
        Info << "Writing new mesh to time " << runTime.name() << endl << endl;
        mesh.write();
    };

    // ---- the loop SM.C:2257-2437 --------------------------------------------------------------------
    // Iterations run in chunks that end at the next write point (SM.C:2416: at the stop, or when (i + 1) % writeInterval == 0
    // and i > 0): one engine call per chunk, no host synchronisation inside it on one rank.
    auto chunkEnd = [&](label i)       // number of iterations from iteration index i up to and including the next write point
    {
        label n = writeInterval - (i % writeInterval);
        if (i == 0 && n == 1) n += writeInterval;      // i = 0 never writes (SM.C:2416 "and (i > 0)")
        return min(centroidalIters - i, n);
    };
    bool stopIteration = false;
    if (!Pstream::parRun())
    {
        for (label i = 0; i < centroidalIters && !stopIteration; )
        {
            const label chunk = chunkEnd(i);
            List<smgpu_iter_stats> st(chunk);
            int32_t done = 0;
            check(smgpu_iterate(h, chunk, relTol, st.begin(), &done));
            for (label k = 0; k < done; ++k)
            {
                runTime++;                                                        // SM.C:2414
                Info<< "Smoothing iteration=" << (i + k + 1) << " nFrozenPoints=" << st[k].nFrozenPoints
                    << " residual=" << st[k].residual << endl;                // SM.C:2396
            }
            i += done;
            if (done > 0 && st[done - 1].residual < relTol)
            {
                Info << "Residual reached relTol, stopping." << endl;
                stopIteration = true;
            }
            if (i == centroidalIters)
            {
                Info << "Maximum centroidalIters reached, stopping." << endl;
                stopIteration = true;
            }
            if (stopIteration || ((i % writeInterval) == 0 && i > 1)) writeMesh();
            if (done == 0) break;
        }
    }
    else
    {
        Exchanger exchange(halo, stream);
        int32_t lDoubles = SMGPU_HALO_L_LAYERS;
        check(smgpu_halo_l_doubles(h, &lDoubles));
        const bool withL = doLayerTreatment || doBoundarySmoothing;
        for (label i = 0; i < centroidalIters; ++i)
        {
            check(smgpu_iter_begin(h));
            if (withL)     // exchange A and the layer / boundary record L leave in ONE group
            {
                exchange({Part{sendA, recvA, SMGPU_HALO_A_DOUBLES * sizeof(double)}, Part{sendL, recvL, size_t(lDoubles) * sizeof(double)}});
            }
            else
            {
                exchange({Part{sendA, recvA, SMGPU_HALO_A_DOUBLES * sizeof(double)}});   // SM.C:134-148, 402-478
            }
            check(smgpu_iter_mid(h));
            exchange({Part{sendF, recvF, sizeof(int32_t)}});                             // SM.C:2374
            check(smgpu_iter_end(h));
            double ls[2];
            checkHip(hipMemcpyAsync(ls, localStats, sizeof(ls), hipMemcpyDeviceToHost, stream));
            checkHip(hipStreamSynchronize(stream));
            check(smgpu_check_error(h));        // an error word raised by a kernel of this iteration (include/smgpu.h)
            scalar residual = ls[0];
            label nFrozenPoints = label(ls[1]);
            reduce(residual, maxOp<scalar>());                                                     // SM.C:1567
            reduce(nFrozenPoints, sumOp<label>());                                                 // SM.C:2396
            Info<< "Smoothing iteration=" << (i + 1) << " nFrozenPoints=" << nFrozenPoints
                << " residual=" << residual << endl;
            if (residual < relTol)                                                                 // SM.C:2401
            {
                Info << "Residual reached relTol, stopping." << endl;
                stopIteration = true;
            }
            if (i == (centroidalIters - 1))
            {
                Info << "Maximum centroidalIters reached, stopping." << endl;
                stopIteration = true;
            }
            runTime++;
            if ((stopIteration) or ((((i + 1) % writeInterval) == 0) and (i > 0))) writeMesh();    // SM.C:2416
            if (stopIteration) break;
        }
    }
    if (Pstream::parRun())
    {
        checkHip(hipStreamSynchronize(stream));
        checkHip(hipFree(sendA)); checkHip(hipFree(recvA)); checkHip(hipFree(sendF)); checkHip(hipFree(recvF));
        checkHip(hipFree(sendL)); checkHip(hipFree(recvL)); checkHip(hipFree(localStats));
    }
    {
        // near-tie census (include/smgpu.h): the engine's acos may differ from glibc's in the last bit; only an angle comparison
        // whose two sides were a few ulp apart could have been decided the other way by the CPU tool
        int64_t nearTies[4] = {0, 0, 0, 0};
        check(smgpu_get_near_ties(h, nearTies));
        label nTies = label(nearTies[0]);
        reduce(nTies, sumOp<label>());
        if (nTies > 0)
            Info<< "WARNING: " << nTies << " angle comparison(s) of this run had their two sides within 4 ulp of each other:"
                << " the CPU smoothMesh may decide such a comparison the other way" << nl << endl;
    }
    check(smgpu_destroy(h));
    Info<< "ClockTime = " << runTime.elapsedClockTime() << " s." << nl << endl;
    Info<< "End" << nl << endl;
    return 0;
}
