/*---------------------------------------------------------------------------*\
  smoothMeshGPU -- OpenFOAM-linked host of the MI355X smoothing engine (libsmgpu)

  COMPILE-UNTESTED IN THIS REPOSITORY: the build image has no OpenFOAM (no wmake, no
  headers).  Written against the OpenFOAM.com v2312-v2506 / OpenFOAM.org 12 API that the
  reference builds on (Allwmake:47, src/Make/options.com, options.org).  What it is: the
  reference's utility with its smoothing loop (src/smoothMesh.C:2257-2437) replaced by
  calls into include/smgpu.h; argList / Time / fvMesh keep doing case and mesh I/O
  (SM.C:1786-1818 createTime / createMesh, SM.C:2416-2431 write).  tests/test_adapter.py
  checks that every smgpu_* entry point used here is declared in include/smgpu.h.

  Covered: the options of the loop (-centroidalIters -relTol -minEdgeLength
  -maxStepLength -relStepFrac -totalMinFreeze -edgeAngleConstraint -faceAngleConstraint
  -minAngle -maxAngle -writeInterval, SM.C:1642-1747, defaults SM.C:1857-1890), serial,
  -layerPatches with its options (serial; SM.C:1749-1775), and -parallel for the loop
  (one rank per GPU; shared-point records travel through Pstream, staged over the host --
  swap exchange() for ncclSend/ncclRecv on device pointers where RCCL is linked).
  Not covered here: boundary point smoothing (constant/geometry/*.obj) and the layer
  set-up under -parallel -- INTEGRATION.md shows the calls; the standalone front-end
  smoothmesh_amd/bin/smoothMesh has all of it.
\*---------------------------------------------------------------------------*/

#include "argList.H"
#include "Time.H"
#include "fvMesh.H"
#include "processorPolyPatch.H"
#include "emptyPolyPatch.H"
#include "labelIOList.H"
#include "PstreamBuffers.H"
#include "PstreamReduceOps.H"
#include "UOPstream.H"
#include "UIPstream.H"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <vector>

#include "smgpu.h"

using namespace Foam;

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

// Shared-point tables of this rank (what smoothmesh_amd/halo.py:HaloTables builds):
// candidates = global ids of the points on my processor patches; a point is shared with
// rank r when r lists it too.  Slots: per peer in ascending rank, ascending global id.
struct HaloTables
{
    std::vector<int32_t> sharedLocal, sendShared, combOffsets, combSlots;
    std::vector<int> peers;
    std::vector<int32_t> counts;     // per peer
    int32_t nSend = 0, nRecv = 0;
};

HaloTables buildHalo(const fvMesh& mesh, const labelList& pointProcAddressing)
{
    HaloTables t;
    const label me = Pstream::myProcNo();
    // my candidates
    labelHashSet mine;
    forAll(mesh.boundaryMesh(), patchI)
    {
        const polyPatch& pp = mesh.boundaryMesh()[patchI];
        if (!isA<processorPolyPatch>(pp)) continue;
        const labelList& mp = pp.meshPoints();
        forAll(mp, i) mine.insert(pointProcAddressing[mp[i]]);
    }
    List<labelList> all(Pstream::nProcs());
    all[me] = mine.sortedToc();
    Pstream::gatherList(all);
    Pstream::scatterList(all);

    std::map<label, label> localOf;      // global id -> local id
    forAll(pointProcAddressing, p) localOf[pointProcAddressing[p]] = p;

    std::map<label, std::vector<int>> sharers;   // global id -> ranks holding it (ascending)
    std::vector<std::vector<label>> with(Pstream::nProcs());
    for (label r = 0; r < Pstream::nProcs(); ++r)
    {
        if (r == me) continue;
        std::set_intersection(all[me].begin(), all[me].end(), all[r].begin(), all[r].end(),
                              std::back_inserter(with[r]));
        for (label g : with[r]) sharers[g].push_back(r);
    }
    std::map<label, int32_t> sharedIndex;
    for (const auto& kv : sharers)
    {
        sharedIndex[kv.first] = static_cast<int32_t>(t.sharedLocal.size());
        t.sharedLocal.push_back(static_cast<int32_t>(localOf[kv.first]));
    }
    std::map<int, int32_t> base;
    for (label r = 0; r < Pstream::nProcs(); ++r)
    {
        if (with[r].empty()) continue;
        t.peers.push_back(r);
        t.counts.push_back(static_cast<int32_t>(with[r].size()));
        base[r] = t.nSend;
        for (label g : with[r]) t.sendShared.push_back(sharedIndex[g]);
        t.nSend += static_cast<int32_t>(with[r].size());
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
            const std::vector<label>& w = with[r];
            const int32_t k = static_cast<int32_t>(std::lower_bound(w.begin(), w.end(), kv.first) - w.begin());
            t.combSlots.push_back(base[r] + k);
        }
        t.combOffsets.push_back(static_cast<int32_t>(t.combSlots.size()));
    }
    return t;
}

// move `bytesPerSlot`-byte records of the send slots to the peers' recv slots (device buffers,
// staged over the host; syncTools::syncPointList's transport, SM.C:134-148,402-478,2374)
void exchange(const HaloTables& t, const void* dSend, void* dRecv, size_t bytesPerSlot, hipStream_t stream,
              std::vector<char>& hSend, std::vector<char>& hRecv)
{
    if (!t.nSend) return;
    hSend.resize(size_t(t.nSend) * bytesPerSlot);
    hRecv.resize(size_t(t.nRecv) * bytesPerSlot);
    checkHip(hipMemcpyAsync(hSend.data(), dSend, hSend.size(), hipMemcpyDeviceToHost, stream));
    checkHip(hipStreamSynchronize(stream));
    PstreamBuffers pBufs(Pstream::commsTypes::nonBlocking);
    size_t off = 0;
    for (size_t i = 0; i < t.peers.size(); ++i)
    {
        UOPstream to(t.peers[i], pBufs);
        to.write(hSend.data() + off, std::streamsize(size_t(t.counts[i]) * bytesPerSlot));
        off += size_t(t.counts[i]) * bytesPerSlot;
    }
    pBufs.finishedSends();
    off = 0;
    for (size_t i = 0; i < t.peers.size(); ++i)
    {
        UIPstream from(t.peers[i], pBufs);
        from.read(hRecv.data() + off, std::streamsize(size_t(t.counts[i]) * bytesPerSlot));
        off += size_t(t.counts[i]) * bytesPerSlot;
    }
    checkHip(hipMemcpyAsync(dRecv, hRecv.data(), hRecv.size(), hipMemcpyHostToDevice, stream));
}

}  // namespace


int main(int argc, char *argv[])
{
    argList::addNote("Centroidal smoothing of a 3-D polyMesh on an AMD MI355X (libsmgpu); options as smoothMesh");
    #include "addRegionOption.H"
    argList::addOption("centroidalIters", "label", "Number of centroidal smoothing iterations (default 1000)");
    argList::addOption("relTol", "scalar", "Relative tolerance for stopping the iterations (default 0.02)");
    argList::addOption("minEdgeLength", "scalar", "Edges shorter than this are not shortened further (default 0.5 * shortest edge)");
    argList::addOption("maxStepLength", "scalar", "Maximum step of a point per iteration (default 0.3 * minEdgeLength)");
    argList::addOption("relStepFrac", "scalar", "Fraction of the centroidal step taken per iteration (default 0.5)");
    argList::addOption("totalMinFreeze", "bool", "Freeze all points of edges shorter than minEdgeLength (default false)");
    argList::addOption("edgeAngleConstraint", "bool", "Prohibit the decrease of small edge-edge angles (default true)");
    argList::addOption("faceAngleConstraint", "bool", "Prohibit the deterioration of face-face angles (default true)");
    argList::addOption("minAngle", "scalar", "Angle (degrees) below which angles may not decrease (default 35)");
    argList::addOption("maxAngle", "scalar", "Angle (degrees) above which face angles may not increase (default 160)");
    argList::addOption("writeInterval", "label", "Write the mesh every this many iterations (default 1000000)");
    argList::addOption("layerPatches", "wordRes", "Patches with boundary layer treatment (serial runs of this host)");
    argList::addOption("layerMaxBlendingFraction", "scalar", "default 0.3");
    argList::addOption("layerEdgeLength", "scalar", "default minEdgeLength");
    argList::addOption("layerExpansionRatio", "scalar", "default 1.3");
    argList::addOption("minLayers", "label", "default 1");
    argList::addOption("maxLayers", "label", "default 4");
    argList::addOption("device", "label", "HIP device ordinal (default: rank modulo the number of devices)");

    #include "setRootCase.H"
    #include "createTime.H"
    #include "createMesh.H"          // SM.C:1814-1818

    if (sizeof(label) != sizeof(int32_t) || sizeof(scalar) != sizeof(double))
    {
        FatalErrorInFunction << "libsmgpu needs WM_LABEL_SIZE=32 and WM_PRECISION_OPTION=DP" << exit(FatalError);
    }

    // ---- hand the mesh over (include/smgpu.h, smgpu_mesh_desc) ----------------------------
    const faceList& faces = mesh.faces();
    labelList faceOffsets(faces.size() + 1, 0);
    forAll(faces, fI) faceOffsets[fI + 1] = faceOffsets[fI] + faces[fI].size();
    labelList facePoints(faceOffsets[faces.size()]);
    forAll(faces, fI) forAll(faces[fI], k) facePoints[faceOffsets[fI] + k] = faces[fI][k];
    const List<unsigned char> isInternalPoint(internalPointMask(mesh));

    int nDev = 0;
    checkHip(hipGetDeviceCount(&nDev));
    const label device = args.getOrDefault<label>("device", Pstream::parRun() ? Pstream::myProcNo() % max(nDev, 1) : 0);

    smgpu_mesh_desc d;
    d.nPoints = mesh.nPoints(); d.nCells = mesh.nCells(); d.nFaces = mesh.nFaces(); d.nInternalFaces = mesh.nInternalFaces();
    d.points = reinterpret_cast<const double*>(mesh.points().cdata());      // Vector<double>: 3 packed doubles
    d.faceOffsets = faceOffsets.cdata(); d.facePoints = facePoints.cdata();
    d.owner = mesh.faceOwner().cdata(); d.neighbour = mesh.faceNeighbour().cdata();
    d.isInternalPoint = isInternalPoint.cdata(); d.isSmoothingSurfacePoint = nullptr;   // BPS.C:404-420 with the features off
    d.device = device; d.stream = nullptr; d.useCallerStream = 0;
    smgpu_handle* h = nullptr;
    check(smgpu_create(&d, &h));

    // ---- parameters, defaults as SM.C:1857-1890 ----------------------------------------------
    double meshMinEdge = 0, meshMaxEdge = 0;
    check(smgpu_mesh_stats(h, &meshMinEdge, &meshMaxEdge));
    reduce(meshMinEdge, minOp<scalar>());                                   // SM.C:1527
    const label centroidalIters = args.getOrDefault<label>("centroidalIters", 1000);
    const scalar relTol = args.getOrDefault<scalar>("relTol", 0.02);
    const scalar minEdgeLength = args.getOrDefault<scalar>("minEdgeLength", 0.5 * meshMinEdge);
    const scalar maxStepLength = args.getOrDefault<scalar>("maxStepLength", 0.3 * minEdgeLength);
    const label writeInterval = args.getOrDefault<label>("writeInterval", 1000000);
    smgpu_params prm;
    prm.maxStepLength = maxStepLength;
    prm.relStepFrac = args.getOrDefault<scalar>("relStepFrac", 0.5);
    prm.minEdgeLength = minEdgeLength;
    prm.totalMinFreeze = args.getOrDefault<bool>("totalMinFreeze", false);
    prm.edgeAngleConstraint = args.getOrDefault<bool>("edgeAngleConstraint", true);
    prm.faceAngleConstraint = args.getOrDefault<bool>("faceAngleConstraint", true);
    prm.minAngle = args.getOrDefault<scalar>("minAngle", 35.0);
    prm.maxAngle = args.getOrDefault<scalar>("maxAngle", 160.0);
    check(smgpu_set_params(h, &prm));
    if (maxStepLength > 0.5 * minEdgeLength)
    {
        WarningInFunction << "maxStepLength is larger than half of minEdgeLength" << endl;
    }

    // ---- boundary layer treatment (serial) ------------------------------------------------------
    if (args.found("layerPatches"))
    {
        if (Pstream::parRun())
        {
            FatalErrorInFunction << "-layerPatches under -parallel: use the step-wise set-up (INTEGRATION.md) or the "
                                 << "standalone front-end" << exit(FatalError);
        }
        const labelHashSet layerIds(mesh.boundaryMesh().patchSet(args.get<wordRes>("layerPatches")));
        labelList pStart(mesh.boundaryMesh().size()), pSize(mesh.boundaryMesh().size());
        List<unsigned char> pKind(mesh.boundaryMesh().size()), pLayer(mesh.boundaryMesh().size());
        forAll(mesh.boundaryMesh(), patchI)
        {
            const polyPatch& pp = mesh.boundaryMesh()[patchI];
            pStart[patchI] = pp.start(); pSize[patchI] = pp.size();
            pKind[patchI] = isA<processorPolyPatch>(pp) ? 1 : isA<emptyPolyPatch>(pp) ? 2 : 0;
            pLayer[patchI] = layerIds.found(patchI) ? 1 : 0;
        }
        smgpu_layer_desc ld;
        ld.nPatches = pStart.size(); ld.patchStart = pStart.cdata(); ld.patchSize = pSize.cdata();
        ld.patchKind = pKind.cdata(); ld.isLayerPatch = pLayer.cdata();
        ld.layerMaxBlendingFraction = args.getOrDefault<scalar>("layerMaxBlendingFraction", 0.3);
        ld.layerEdgeLength = args.getOrDefault<scalar>("layerEdgeLength", minEdgeLength);
        ld.layerExpansionRatio = args.getOrDefault<scalar>("layerExpansionRatio", 1.3);
        ld.minLayers = args.getOrDefault<label>("minLayers", 1);
        ld.maxLayers = args.getOrDefault<label>("maxLayers", 4);
        int32_t doLayerTreatment = 0;
        check(smgpu_set_layers(h, &ld, &doLayerTreatment));
        Info<< "Boundary layer treatment " << (doLayerTreatment ? "enabled" : "disabled") << endl;
    }

    auto writeMesh = [&]()
    {
        pointField newPoints(mesh.nPoints());
        check(smgpu_get_points(h, reinterpret_cast<double*>(newPoints.data())));
        mesh.movePoints(newPoints);          // OpenFOAM's own geometry is only needed for the write
        IOstream::defaultPrecision(max(10u, IOstream::defaultPrecision()));   // SM.C:2425
        mesh.write();
    };

    if (!Pstream::parRun())
    {
        // ---- the loop SM.C:2257-2437 on one rank -------------------------------------------------
        for (label i = 0; i < centroidalIters; )
        {
            const label chunk = min(centroidalIters - i, writeInterval - (i % writeInterval));
            List<smgpu_iter_stats> st(chunk);
            int32_t done = 0;
            check(smgpu_iterate(h, chunk, relTol, st.data(), &done));
            for (label k = 0; k < done; ++k)
            {
                runTime++;
                Info<< "Smoothing iteration=" << (i + k + 1) << " nFrozenPoints=" << st[k].nFrozenPoints
                    << " residual=" << st[k].residual << endl;                // SM.C:2396
            }
            i += done;
            const bool stop = (done > 0 && st[done - 1].residual < relTol) || i >= centroidalIters || done < chunk;
            if (stop || (i % writeInterval) == 0) writeMesh();
            if (stop) break;
        }
    }
    else
    {
        // ---- -parallel: one rank per GPU, shared-point exchange between the calls -----------------
        labelIOList pointProcAddressing
        (
            IOobject("pointProcAddressing", mesh.facesInstance(), polyMesh::meshSubDir, mesh,
                     IOobject::MUST_READ, IOobject::NO_WRITE)
        );
        const HaloTables t(buildHalo(mesh, pointProcAddressing));
        void *sendA = nullptr, *recvA = nullptr, *sendF = nullptr, *recvF = nullptr, *localStats = nullptr;
        checkHip(hipSetDevice(device));
        checkHip(hipMalloc(&sendA, size_t(max(t.nSend, 1)) * SMGPU_HALO_A_DOUBLES * sizeof(double)));
        checkHip(hipMalloc(&recvA, size_t(max(t.nRecv, 1)) * SMGPU_HALO_A_DOUBLES * sizeof(double)));
        checkHip(hipMalloc(&sendF, size_t(max(t.nSend, 1)) * sizeof(int32_t)));
        checkHip(hipMalloc(&recvF, size_t(max(t.nRecv, 1)) * sizeof(int32_t)));
        checkHip(hipMalloc(&localStats, 2 * sizeof(double)));
        smgpu_halo_desc hd;
        hd.nShared = int32_t(t.sharedLocal.size()); hd.sharedLocal = t.sharedLocal.data();
        hd.nSend = t.nSend; hd.sendShared = t.sendShared.data();
        hd.nRecv = t.nRecv; hd.combOffsets = t.combOffsets.data(); hd.combSlots = t.combSlots.data();
        hd.sendA = sendA; hd.recvA = recvA; hd.sendF = sendF; hd.recvF = recvF; hd.localStats = localStats;
        hd.sendL = nullptr; hd.recvL = nullptr;
        hd.useExchangeStream = 0; hd.exchangeStream = nullptr;      // exchanges in order on the engine's stream
        check(smgpu_halo_configure(h, &hd));
        void* vs = nullptr;
        check(smgpu_get_stream(h, &vs));
        hipStream_t stream = static_cast<hipStream_t>(vs);
        std::vector<char> hs, hr;
        for (label i = 0; i < centroidalIters; ++i)
        {
            check(smgpu_iter_begin(h));
            exchange(t, sendA, recvA, SMGPU_HALO_A_DOUBLES * sizeof(double), stream, hs, hr);   // SM.C:134-148, 402-478
            check(smgpu_iter_mid(h));
            exchange(t, sendF, recvF, sizeof(int32_t), stream, hs, hr);                            // SM.C:2374
            check(smgpu_iter_end(h));
            double ls[2];
            checkHip(hipMemcpyAsync(ls, localStats, sizeof(ls), hipMemcpyDeviceToHost, stream));
            checkHip(hipStreamSynchronize(stream));
            scalar residual = ls[0];
            label nFrozenPoints = label(ls[1]);
            reduce(residual, maxOp<scalar>());                                                     // SM.C:1567
            reduce(nFrozenPoints, sumOp<label>());                                                 // SM.C:2396
            runTime++;
            Info<< "Smoothing iteration=" << (i + 1) << " nFrozenPoints=" << nFrozenPoints
                << " residual=" << residual << endl;
            const bool stop = residual < relTol || i + 1 == centroidalIters;                       // SM.C:2401
            if (stop || ((i + 1) % writeInterval) == 0) writeMesh();
            if (stop) break;
        }
        checkHip(hipFree(sendA)); checkHip(hipFree(recvA)); checkHip(hipFree(sendF)); checkHip(hipFree(recvF));
        checkHip(hipFree(localStats));
    }
    check(smgpu_destroy(h));
    Info<< "End" << nl << endl;
    return 0;
}
