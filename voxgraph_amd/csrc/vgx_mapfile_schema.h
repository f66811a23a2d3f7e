// Field numbers of the protobuf messages inside voxblox / cblox map files.  [recalled]: the .proto
// files live in un-vendored dependencies (voxblox/proto/voxblox/{Layer,Block}.proto,
// cblox/proto/cblox/{SubmapCollection,Submap,QuatTransformation}.proto) and no real file was
// available to check against; everything the reader/writer assumes about them is in this table.
#ifndef VOXGRAPH_AMD_CSRC_VGX_MAPFILE_SCHEMA_H_
#define VOXGRAPH_AMD_CSRC_VGX_MAPFILE_SCHEMA_H_
namespace vgx_schema {
// voxblox LayerProto
constexpr int kLayerVoxelSize = 1;      // double
constexpr int kLayerVoxelsPerSide = 2;  // uint32
constexpr int kLayerType = 3;           // string: "tsdf" | "esdf" | ...
// voxblox BlockProto
constexpr int kBlockVoxelsPerSide = 1;  // int32
constexpr int kBlockVoxelSize = 2;      // double
constexpr int kBlockOriginX = 3;        // double
constexpr int kBlockOriginY = 4;
constexpr int kBlockOriginZ = 5;
constexpr int kBlockHasData = 6;        // bool
constexpr int kBlockVoxelData = 7;      // repeated uint32
// Block<TsdfVoxel>::serializeToIntegers: 3 words per voxel = distance bits, weight bits,
// a | b << 8 | g << 16 | r << 24
constexpr int kTsdfWordsPerVoxel = 3;
// Block<EsdfVoxel>::serializeToIntegers: 2 words per voxel = distance bits, then
// parent.x << 24 | parent.y << 16 | parent.z << 8 | observed (int8 parent components; only the low
// byte, `observed`, is read here; the writer emits zero parents -- REG never uses them and voxblox
// recomputes them when it propagates)
constexpr int kEsdfWordsPerVoxel = 2;
// cblox SubmapCollectionProto
constexpr int kCollectionVoxelSize = 1;      // double
constexpr int kCollectionVoxelsPerSide = 2;  // uint32
constexpr int kCollectionNumSubmaps = 3;     // uint32
// cblox SubmapProto
constexpr int kSubmapId = 1;             // uint64
constexpr int kSubmapNumBlocks = 2;      // uint32 (TSDF)
constexpr int kSubmapTransform = 3;      // QuatTransformationProto
constexpr int kSubmapNumEsdfBlocks = 4;  // uint32
// cblox QuatTransformationProto { PositionProto position = 1; QuaternionProto rotation = 2; }
constexpr int kTransformPosition = 1;    // message {double x = 1, y = 2, z = 3}
constexpr int kTransformRotation = 2;    // message {double w = 1, x = 2, y = 3, z = 4}
}  // namespace vgx_schema
#endif
