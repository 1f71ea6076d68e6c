// polymesh_io.hpp -- OpenFOAM polyMesh directory reader / writer (host, no GPU).
// Stands in for createMesh.H (src/smoothMesh.C:1814-1818) and mesh.write() (SM.C:2416-2431);
// file formats per SURVEY Appendix C (no polyMesh sample ships with the reference).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace smhost {

struct PatchInfo {
    std::string name, type;
    int32_t nFaces = 0, startFace = 0;
    int32_t myProcNo = -1, neighbProcNo = -1;
};

struct PolyMeshData {
    std::vector<double> points;       // 3 * nPoints
    std::vector<int32_t> faceOffsets; // nFaces + 1
    std::vector<int32_t> facePoints;
    std::vector<int32_t> owner;       // nFaces
    std::vector<int32_t> neighbour;   // nInternalFaces
    std::vector<PatchInfo> patches;
    int32_t nCells = 0;
    int32_t nPoints() const { return (int32_t)(points.size() / 3); }
    int32_t nFaces() const { return (int32_t)owner.size(); }
    int32_t nInternalFaces() const { return (int32_t)neighbour.size(); }
};

// All functions throw std::runtime_error with a message naming the file.
void readPolyMesh(const std::string& polyMeshDir, const std::string& pointsDir, PolyMeshData& out);
void readPoints(const std::string& file, std::vector<double>& pts);
void readLabelList(const std::string& file, std::vector<int32_t>& out);
// Wavefront OBJ inputs of the boundary point smoothing (constant/geometry/*.obj, SM.C:1924-1926).  As OpenFOAM's readers
// (third-party) treat them: surface polygons become triangle fans about their first vertex; an edge mesh takes the
// consecutive pairs of every "l" record and drops the points no edge uses, keeping the order of the others.
void readObjSurface(const std::string& file, std::vector<double>& points, std::vector<int32_t>& triangles);
void readObjEdges(const std::string& file, std::vector<double>& points, std::vector<int32_t>& edges);
void writePoints(const std::string& polyMeshDir, const std::string& location, int32_t nPoints, const double* pts,
                 bool binary, int precision);
void writeLabelList(const std::string& file, const std::string& location, const std::string& object,
                    const std::string& cls, int64_t n, const int32_t* v, bool binary, const std::string& note = "");
void writePolyMesh(const std::string& polyMeshDir, const std::string& location, const PolyMeshData& m, bool binary,
                   int precision);
// OpenFOAM's writeCompression: every file written afterwards becomes <file>.gz (reading accepts both always)
void setWriteCompression(bool on);
void makeDirs(const std::string& path);
bool fileExists(const std::string& path);
bool dirExists(const std::string& path);

}  // namespace smhost
