// Stand-in for cblox::SubmapCollection<SubmapType> ([recalled]): an id -> shared_ptr map with the accessors voxgraph's
// backend and tools call (getSubmapConstPtr, getIDs, size); addSubmap() is how the check drivers fill it.
// TEST INFRASTRUCTURE -- see oracle/ref_shims/README.md.
#ifndef ORACLE_REF_SHIMS_CBLOX_CORE_SUBMAP_COLLECTION_H_
#define ORACLE_REF_SHIMS_CBLOX_CORE_SUBMAP_COLLECTION_H_
#include <map>
#include <memory>
#include <vector>

#include "cblox/core/common.h"
#include "cblox/core/tsdf_esdf_submap.h"
namespace cblox {
template <typename SubmapType>
class SubmapCollection {
 public:
  typedef std::shared_ptr<SubmapCollection> Ptr;
  typedef std::shared_ptr<const SubmapCollection> ConstPtr;
  void addSubmap(const std::shared_ptr<SubmapType>& submap) { submaps_[submap->getID()] = submap; }
  std::shared_ptr<const SubmapType> getSubmapConstPtr(const SubmapID id) const {
    auto it = submaps_.find(id);
    return it == submaps_.end() ? nullptr : it->second;
  }
  std::vector<SubmapID> getIDs() const {
    std::vector<SubmapID> ids;
    for (const auto& kv : submaps_) ids.push_back(kv.first);
    return ids;
  }
  size_t size() const { return submaps_.size(); }

 private:
  std::map<SubmapID, std::shared_ptr<SubmapType> > submaps_;
};
}  // namespace cblox
#endif
