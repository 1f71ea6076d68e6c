/* smhost.h -- C-ABI of the host side either side of the smoothing loop: OpenFOAM polyMesh
 * directory I/O (what createMesh.H does at src/smoothMesh.C:1814-1818 and mesh.write() at
 * :2416-2431).  Plain C++ (no GPU); used by the bundled `smoothMesh` front-end and, through
 * ctypes, by the Python tests and mesh generators.
 *
 * Formats: FoamFile header + ascii or binary payload (faceList / faceCompactList, labelList,
 * vectorField, polyBoundaryMesh), label = 32 or 64 bit on read (header `arch`), label=32 /
 * scalar=64 on write.  <file>.gz is read wherever <file> is missing (zlib), as OpenFOAM does.
 * All functions return 0 on success; message through smhost_last_error().
 */
#ifndef SMHOST_H
#define SMHOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct smhost_mesh smhost_mesh;

const char* smhost_last_error(void);

/* Read <dir>/{points,faces,owner,neighbour,boundary}; pointsDir (may be NULL) overrides where
 * `points` is read from (a later time directory's polyMesh, like OpenFOAM's pointsInstance). */
int smhost_read_polymesh(const char* polyMeshDir, const char* pointsDir, smhost_mesh** out);
void smhost_mesh_free(smhost_mesh* m);
int smhost_mesh_sizes(const smhost_mesh* m, int32_t* nPoints, int32_t* nCells, int32_t* nFaces,
                      int32_t* nInternalFaces, int32_t* nPatches, int64_t* nnzFacePoints);
int smhost_mesh_copy(const smhost_mesh* m, double* points, int32_t* faceOffsets, int32_t* facePoints,
                     int32_t* owner, int32_t* neighbour);
int smhost_mesh_patch(const smhost_mesh* m, int32_t i, char* name, int32_t nameCap, char* type, int32_t typeCap,
                      int32_t* nFaces, int32_t* startFace, int32_t* myProcNo, int32_t* neighbProcNo);

/* Write a complete polyMesh directory (mesh generators, decomposition). */
int smhost_write_polymesh(const char* polyMeshDir, const char* location, int32_t nPoints, const double* points,
                          int32_t nFaces, const int32_t* faceOffsets, const int32_t* facePoints, const int32_t* owner,
                          int32_t nInternalFaces, const int32_t* neighbour, int32_t nCells, int32_t nPatches,
                          const char* const* patchNames, const char* const* patchTypes, const int32_t* patchNFaces,
                          const int32_t* patchStart, const int32_t* patchMyProc, const int32_t* patchNbrProc,
                          int32_t binary, int32_t precision);
/* controlDict's writeCompression: files written afterwards are gzip-compressed (<file>.gz).  Reading accepts a
 * <file>.gz wherever <file> is missing, as OpenFOAM does. */
int smhost_set_write_compression(int32_t on);
/* Write <dir>/points only (mesh.write() of a moved mesh, SM.C:2430); precision as SM.C:2425. */
int smhost_write_points(const char* polyMeshDir, const char* location, int32_t nPoints, const double* points,
                        int32_t binary, int32_t precision);
/* labelList file (pointProcAddressing etc.): n = -1 on input to query the size into *n. */
int smhost_read_label_list(const char* file, int32_t* out, int64_t* n);
/* points file alone (vectorField; a later time directory's polyMesh/points, SM.C:2430 written, read back on restart):
 * *n = number of scalars (3 per point); out == NULL or *n too small: only the size is returned. */
int smhost_read_points(const char* file, double* out, int64_t* n);
/* Wavefront OBJ inputs of the boundary point smoothing (constant/geometry/ .obj files, SM.C:1924-1926), read the way
 * OpenFOAM's readers do: kind 0 = surface (triSurface: polygons as triangle fans about their first vertex; 3 ids per
 * element), kind 1 = edge mesh (edgeMesh: consecutive pairs of every "l" record, unused points dropped; 2 ids per
 * element).  Two calls: with points == NULL only the counts are returned. */
int smhost_read_obj(const char* file, int32_t kind, double* points, int64_t* nPoints, int32_t* elements, int64_t* nElements);
int smhost_write_label_list(const char* file, const char* location, const char* object, const char* cls,
                            int64_t n, const int32_t* values, int32_t binary);

/* Synthetic polyhedral mesh (stands in for snappyHexMesh, BASELINE configs[3-4]): castellated one-level
 * octree mesh of the unit cube with a spherical cavity; see csrc/host/meshgen.cpp. */
int smhost_gen_cavity_mesh(int32_t N, double radius, double shell, double jitter, uint64_t seed, smhost_mesh** out);
/* Sub-domain `rank` of the (px, py, pz) box decomposition of that mesh, generated directly in decomposePar layout
 * (processor patches, local numbering in ascending global order: what processorN/constant/polyMesh would hold; the
 * reference reads exactly that per rank, testcase/run_parallel:11-19).  Only the box is visited, so no process ever
 * holds the global mesh.  Coarse cell i of an axis goes to box b with floor(b N / P) <= i < floor((b+1) N / P). */
int smhost_gen_cavity_subdomain(int32_t N, double radius, double shell, double jitter, uint64_t seed, int32_t px, int32_t py,
                                int32_t pz, int32_t rank, smhost_mesh** out);
/* Global ids of a smhost_gen_cavity_subdomain mesh (either pointer may be NULL): per point its index in the global
 * (2N+1)^3 lattice (unique per global point and ascending with the global point id: a pointProcAddressing for the
 * shared-point tables), per cell 8 * coarse cell index + child (ascending with the global cell id). */
int smhost_mesh_global_ids(const smhost_mesh* m, int64_t* pointGlobal, int64_t* cellGlobal);

#ifdef __cplusplus
}
#endif
#endif
