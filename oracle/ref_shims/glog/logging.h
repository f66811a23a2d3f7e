// CHECK* macros standing in for glog.  TEST INFRASTRUCTURE (oracle/ref_shims/README.md).
#ifndef ORACLE_REF_SHIMS_GLOG_LOGGING_H_
#define ORACLE_REF_SHIMS_GLOG_LOGGING_H_
#include <cstdio>
#include <cstdlib>
#include <sstream>
namespace ref_shims {
class CheckFailure {
 public:
  CheckFailure(const char* file, int line, const char* what) {
    s_ << file << ":" << line << " CHECK failed: " << what << " ";
  }
  [[noreturn]] ~CheckFailure() {
    std::fprintf(stderr, "%s\n", s_.str().c_str());
    std::abort();
  }
  std::ostream& stream() { return s_; }

 private:
  std::ostringstream s_;
};
}  // namespace ref_shims
#define CHECK(cond) \
  if (cond) {       \
  } else            \
    ::ref_shims::CheckFailure(__FILE__, __LINE__, #cond).stream()
#define CHECK_EQ(a, b) CHECK((a) == (b))
#define CHECK_NE(a, b) CHECK((a) != (b))
#define CHECK_LE(a, b) CHECK((a) <= (b))
#define CHECK_LT(a, b) CHECK((a) < (b))
#define CHECK_GE(a, b) CHECK((a) >= (b))
#define CHECK_GT(a, b) CHECK((a) > (b))
#define CHECK_NOTNULL(p) (p)
// LOG(FATAL) << ...: abort with the message (the only severity voxgraph's backend uses on these paths)
#define VGX_SHIM_LOG_FATAL ::ref_shims::CheckFailure(__FILE__, __LINE__, "LOG(FATAL)").stream()
#define LOG(severity) VGX_SHIM_LOG_##severity
#define CHECK_NEAR(a, b, margin) CHECK(((a) - (b)) <= (margin) && ((b) - (a)) <= (margin))
#endif
