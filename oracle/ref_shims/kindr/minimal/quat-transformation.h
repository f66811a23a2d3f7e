// Stand-in for minkindr's QuatTransformationTemplate ([recalled]).
// TEST INFRASTRUCTURE -- see oracle/ref_shims/README.md.
#ifndef ORACLE_REF_SHIMS_KINDR_MINIMAL_QUAT_TRANSFORMATION_H_
#define ORACLE_REF_SHIMS_KINDR_MINIMAL_QUAT_TRANSFORMATION_H_
#include <cmath>
#include <limits>

#include <Eigen/Core>

namespace kindr {
namespace minimal {

template <typename Scalar>
class RotationQuaternionTemplate {
 public:
  typedef Eigen::Matrix<Scalar, 3, 1> Vector3;
  RotationQuaternionTemplate() : w_(1), v_() {}
  RotationQuaternionTemplate(Scalar w, Scalar x, Scalar y, Scalar z) : w_(w), v_(x, y, z) {}
  Scalar w() const { return w_; }
  Scalar x() const { return v_[0]; }
  Scalar y() const { return v_[1]; }
  Scalar z() const { return v_[2]; }
  const Vector3& imaginary() const { return v_; }

  // exponential map of an angle-axis vector (Grassia 1998); the half-angle
  // trigonometry is carried out in double and narrowed on construction
  static RotationQuaternionTemplate exp(const Vector3& dx) {
    const double theta = static_cast<double>(dx.norm());
    double na;
    if (theta < std::pow(std::numeric_limits<double>::epsilon(), 0.25)) {
      na = 0.5 + (theta * theta) * (1.0 / 48.0);
    } else {
      na = std::sin(theta * 0.5) / theta;
    }
    const double ct = std::cos(theta * 0.5);
    return RotationQuaternionTemplate(
        static_cast<Scalar>(ct), static_cast<Scalar>(static_cast<double>(dx[0]) * na),
        static_cast<Scalar>(static_cast<double>(dx[1]) * na),
        static_cast<Scalar>(static_cast<double>(dx[2]) * na));
  }
  Vector3 log() const {
    const Scalar na = v_.norm();
    const Scalar eta = w_;
    Scalar scale;
    if (na < std::numeric_limits<Scalar>::epsilon()) {
      scale = Scalar(1) / (eta == Scalar(0) ? Scalar(1) : eta);
    } else if (std::fabs(eta) < na) {
      scale = (eta >= 0 ? std::acos(eta) : -std::acos(-eta)) / na;
    } else {
      scale = (eta > 0 ? std::asin(na) : -std::asin(na)) / na;
    }
    return v_ * (Scalar(2) * scale);
  }
  RotationQuaternionTemplate inverse() const { return RotationQuaternionTemplate(w_, -v_[0], -v_[1], -v_[2]); }
  // Eigen's generic quaternion product
  RotationQuaternionTemplate operator*(const RotationQuaternionTemplate& b) const {
    const RotationQuaternionTemplate& a = *this;
    return RotationQuaternionTemplate(
        a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
        a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
        a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
        a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
  }
  // Eigen's QuaternionBase::_transformVector: uv = 2 (u x v);  v + w uv + u x uv
  Vector3 rotate(const Vector3& v) const {
    Vector3 uv = v_.cross(v);
    uv = uv + uv;
    const Vector3 c = v_.cross(uv);
    Vector3 out;
    for (int i = 0; i < 3; ++i) out[i] = v[i] + w_ * uv[i] + c[i];
    return out;
  }

 private:
  Scalar w_;
  Vector3 v_;
};

template <typename Scalar>
class QuatTransformationTemplate {
 public:
  typedef Eigen::Matrix<Scalar, 3, 1> Position;
  typedef Eigen::Matrix<Scalar, 3, 1> Vector3;
  typedef Eigen::Matrix<Scalar, 6, 1> Vector6;
  typedef RotationQuaternionTemplate<Scalar> Rotation;

  QuatTransformationTemplate() {}
  QuatTransformationTemplate(const Rotation& q, const Position& t) : q_(q), t_(t) {}

  // translation is copied, not multiplied by the SO(3) left Jacobian
  static QuatTransformationTemplate exp(const Vector6& v) {
    return QuatTransformationTemplate(Rotation::exp(v.template tail<3>()), v.template head<3>());
  }
  Vector6 log() const {
    const Position a = q_.log();
    Vector6 v;
    for (int i = 0; i < 3; ++i) {
      v[i] = t_[i];
      v[3 + i] = a[i];
    }
    return v;
  }
  const Rotation& getRotation() const { return q_; }
  Eigen::Matrix<Scalar, 4, 4> getTransformationMatrix() const {
    Eigen::Matrix<Scalar, 4, 4> m;
    for (int c = 0; c < 3; ++c) {
      Position e;
      e[c] = Scalar(1);
      const Position col = q_.rotate(e);
      for (int r = 0; r < 3; ++r) m(r, c) = col[r];
      m(c, 3) = t_[c];
    }
    m(3, 3) = Scalar(1);
    return m;
  }
  const Position& getPosition() const { return t_; }
  QuatTransformationTemplate inverse() const {
    const Rotation qi = q_.inverse();
    return QuatTransformationTemplate(qi, -qi.rotate(t_));
  }
  QuatTransformationTemplate operator*(const QuatTransformationTemplate& rhs) const {
    return QuatTransformationTemplate(q_ * rhs.q_, t_ + q_.rotate(rhs.t_));
  }
  Position operator*(const Position& p) const { return q_.rotate(p) + t_; }

 private:
  Rotation q_;
  Position t_;
};

}  // namespace minimal
}  // namespace kindr
#endif
