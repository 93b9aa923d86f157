// wg_abstract_robot.hh -- the robot-model interface of the reference's public API, in the shape of abstract-robot-dynamics
// (CjrlHumanoidDynamicRobot / CjrlFoot / CjrlJoint, >= 1.15; NOT vendored by the reference and absent from this image) and
// the jrl-mal vector / matrix spellings (MAL_*) the reference's public header and its callers use.
//
// Only what the Herdt-2010 / Kajita stage-1 path calls is declared -- each method cites its call site in the reference.
// A caller that owns a real abstract-robot-dynamics robot wraps it in a ten-line adapter deriving from these classes
// (INTEGRATION.md shows it); a caller without one (tests, fleets of simulated robots) derives them directly, e.g. from a
// table of link transforms.  The generator itself reads ~14 numbers from the model -- see
// HumanoidModelFromRobot in wg_walkgen.hh -- plus forward kinematics of the start posture for EvaluateStartingState.
#ifndef WG_ABSTRACT_ROBOT_HH
#define WG_ABSTRACT_ROBOT_HH

#include <cstddef>
#include <string>
#include <vector>

// ---- jrl-mal spellings (include/jrl/walkgen/patterngeneratorinterface.hh uses MAL_VECTOR, MAL_VECTOR_TYPE, MAL_S3_VECTOR,
// MAL_S3_VECTOR_TYPE, MAL_S4x4_MATRIX; the reference's tests add MAL_VECTOR_DIM / _RESIZE / _SIZE / _FILL) -----------------
struct vectorN : public std::vector<double> {          // ublas::vector<double>: operator() and operator[] both index
  vectorN() {}
  explicit vectorN(size_t n, double v = 0.0) : std::vector<double>(n, v) {}
  vectorN(const std::vector<double> &o) : std::vector<double>(o) {}
  double &operator()(size_t i) { return (*this)[i]; }
  const double &operator()(size_t i) const { return (*this)[i]; }
};
struct vector3d {
  double v[3];
  vector3d() { v[0] = v[1] = v[2] = 0.0; }
  vector3d(double x, double y, double z) { v[0] = x; v[1] = y; v[2] = z; }
  double &operator()(size_t i) { return v[i]; }
  const double &operator()(size_t i) const { return v[i]; }
  double &operator[](size_t i) { return v[i]; }
  const double &operator[](size_t i) const { return v[i]; }
};
struct matrix4d {                                       // row-major 4 x 4, identity by default
  double m[16];
  matrix4d() { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0 : 0.0; }
  double &operator()(size_t i, size_t j) { return m[4 * i + j]; }
  const double &operator()(size_t i, size_t j) const { return m[4 * i + j]; }
};
#ifndef MAL_VECTOR_TYPE
#define MAL_VECTOR_TYPE(type) vectorN
#define MAL_VECTOR(name, type) vectorN name
#define MAL_VECTOR_DIM(name, type, nb_rows) vectorN name(nb_rows)
#define MAL_VECTOR_SIZE(name) name.size()
#define MAL_VECTOR_RESIZE(name, nb_rows) name.resize(nb_rows)
#define MAL_VECTOR_FILL(name, value) name.assign(name.size(), value)
#define MAL_S3_VECTOR_TYPE(type) vector3d
#define MAL_S3_VECTOR(name, type) vector3d name
#define MAL_S3_VECTOR_ACCESS(name, i) name[i]
#define MAL_S4x4_MATRIX_TYPE(type) matrix4d
#define MAL_S4x4_MATRIX(name, type) matrix4d name
#define MAL_S4x4_MATRIX_ACCESS_I_J(name, i, j) name(i, j)
#endif

// ---- abstract-robot-dynamics, the methods on this path ------------------------------------------------------------------
class CjrlJoint {
 public:
  virtual ~CjrlJoint() {}
  // hip-yaw limits: OrientationsPreview.cpp:49-68 (radians, rad/s; equal bounds mean "not given")
  virtual double lowerBound(unsigned int inDofRank) const = 0;
  virtual double upperBound(unsigned int inDofRank) const = 0;
  virtual double upperVelocityBound(unsigned int inDofRank) const = 0;
  // pose of the joint frame in the world for the current configuration, and in the robot's reference posture:
  // ComAndFootRealizationByGeometry::InitializationFoot, ComAndFootRealizationByGeometry.cpp:384-440
  virtual const matrix4d &currentTransformation() const = 0;
  virtual const matrix4d &initialPosition() const = 0;
};

class CjrlFoot {
 public:
  virtual ~CjrlFoot() {}
  virtual const CjrlJoint *associatedAnkle() const = 0;                         // OrientationsPreview.cpp:47, 62
  virtual void getSoleSize(double &outLength, double &outWidth) const = 0;      // relative-feet-inequalities.cpp:163, 171
  virtual void getAnklePositionInLocalFrame(vector3d &outCoordinates) const = 0;   // rigid-body-system.cpp:38, 176
};

class CjrlHumanoidDynamicRobot {
 public:
  virtual ~CjrlHumanoidDynamicRobot() {}
  virtual double mass() const = 0;                                              // ZMPVelocityReferencedQP.cpp:68
  virtual CjrlFoot *leftFoot() const = 0;
  virtual CjrlFoot *rightFoot() const = 0;
  virtual CjrlJoint *waist() const = 0;                                         // OrientationsPreview.cpp:44
  // joints on the chain between two joints, ends included; [1] is the hip-yaw joint (OrientationsPreview.cpp:48, 63)
  virtual std::vector<CjrlJoint *> jointsBetween(const CjrlJoint &inStartJoint, const CjrlJoint &inEndJoint) const = 0;
  virtual unsigned int numberDof() const = 0;                                   // tests/TestObject.cpp:176
  virtual std::vector<CjrlJoint *> getActuatedJoints() const = 0;               // tests/TestObject.cpp:179
  virtual bool setProperty(std::string &inProperty, const std::string &inValue) = 0;   // ..ByGeometry.cpp:358-363
  // forward kinematics of the start posture: free flyer (x y z roll pitch yaw) followed by the joint values
  virtual bool currentConfiguration(const vectorN &inConfig) = 0;               // ..ByGeometry.cpp:352
  virtual const vectorN &currentConfiguration() const = 0;                      // ..ByGeometry.cpp:368
  virtual bool computeForwardKinematics() = 0;                                  // ..ByGeometry.cpp:366
  virtual const vector3d &positionCenterOfMass() const = 0;                     // ..ByGeometry.cpp:497
};

#endif
