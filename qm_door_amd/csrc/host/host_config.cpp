// Host-side configuration of the hot path: URDF -> flat 24-DoF model, task/reference/gait .info -> settings.
//
// Mirrors (reference tree):
//   qm_interface/src/QMInterface.cpp:37-74, 79-142      QMInterface ctor / setupOptimalControlProblem (what is read, from where)
//   qm_interface/src/common/ModelSettings.cpp:15-41      model_settings
//   qm_interface/include/qm_interface/common/ModelSettings.h:32-38   joint / contact name lists
//   qm_interface/src/QMInterface.cpp:408-416             createPinocchioInterface(urdf, jointNames): joints not listed are fixed
//   qm_interface/src/QMInterface.cpp:455-480             gait schedule loading
//   qm_wbc/src/WbcBase.cpp:597-627                       loadTasksSetting (effort limits, WBC friction coefficient)
//   qm_wbc/cfg/wbcWigeht.cfg:7-47                        default WBC gains
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <set>

#include "../../../include/qmgpu.h"
#include "host_error.h"
#include "text_formats.h"

namespace qmhost {

thread_local std::string g_lastError;

// Canonical joint order = Pinocchio's order for this URDF (children visited alphabetically by joint name,
// which is how urdfdom fills child_joints) -- SURVEY.md Appendix D.
static const char* kJointNames[QMGPU_NJ] = {"LF_HAA", "LF_HFE", "LF_KFE", "LH_HAA", "LH_HFE", "LH_KFE", "RF_HAA", "RF_HFE", "RF_KFE",
                                            "RH_HAA", "RH_HFE", "RH_KFE", "z1_joint_1", "z1_joint_2", "z1_joint_3", "z1_joint_4", "z1_joint_5", "z1_joint_6"};
static const char* kContactNames[QMGPU_NC] = {"LF_FOOT", "RF_FOOT", "LH_FOOT", "RH_FOOT"};  // ModelSettings.h:36

struct UrdfLink {
  std::string name;
  double mass = 0, com[3] = {0, 0, 0}, I[6] = {0, 0, 0, 0, 0, 0};
};
struct UrdfJoint {
  std::string name, type, parent, child;
  double xyz[3] = {0, 0, 0}, rpy[3] = {0, 0, 0}, axis[3] = {1, 0, 0};
  double lower = 0, upper = 0, effort = 0, velocity = 0;
};

// rigid-body inertia accumulation: (m, c, I about c) all expressed in one frame
struct Inertia {
  double m = 0, c[3] = {0, 0, 0}, I[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  void add(double m2, const double c2[3], const double I2[3][3]) {
    if (m2 == 0.0 && m == 0.0) return;
    const double mt = m + m2;
    double ct[3];
    for (int i = 0; i < 3; ++i) ct[i] = (m * c[i] + m2 * c2[i]) / mt;
    double It[3][3];
    auto shift = [&](double mm, const double cc[3], const double II[3][3]) {
      const double d[3] = {cc[0] - ct[0], cc[1] - ct[1], cc[2] - ct[2]};
      const double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) It[i][j] += II[i][j] + mm * ((i == j ? d2 : 0.0) - d[i] * d[j]);
    };
    for (auto& r : It) for (auto& e : r) e = 0.0;
    shift(m, c, I);
    shift(m2, c2, I2);
    m = mt;
    for (int i = 0; i < 3; ++i) { c[i] = ct[i]; for (int j = 0; j < 3; ++j) I[i][j] = It[i][j]; }
  }
};

static void buildModel(const std::string& urdfFile, const std::string& eeFrame, qmgpu_model& md) {
  auto doc = XmlParser::parseFile(urdfFile);
  const XmlNode* robot = doc->child("robot");
  if (!robot) throw std::runtime_error("URDF: no <robot> element in " + urdfFile);
  std::map<std::string, UrdfLink> links;
  std::map<std::string, UrdfJoint> joints;  // std::map => alphabetical, as in urdfdom
  for (const auto& n : robot->children) {
    if (n->name == "link") {
      UrdfLink l;
      l.name = n->get("name");
      if (const XmlNode* in = n->child("inertial")) {
        if (const XmlNode* o = in->child("origin")) {
          parseTriple(o->get("xyz", "0 0 0"), l.com, "inertial origin");
          double rpy[3];
          parseTriple(o->get("rpy", "0 0 0"), rpy, "inertial rpy");
          if (rpy[0] != 0 || rpy[1] != 0 || rpy[2] != 0) throw UnsupportedModel("inertial frame of link " + l.name + " is rotated");
        }
        if (const XmlNode* m = in->child("mass")) l.mass = std::strtod(m->get("value", "0").c_str(), nullptr);
        if (const XmlNode* i = in->child("inertia")) {
          const char* k[6] = {"ixx", "ixy", "ixz", "iyy", "iyz", "izz"};
          for (int a = 0; a < 6; ++a) l.I[a] = std::strtod(i->get(k[a], "0").c_str(), nullptr);
        }
      }
      links[l.name] = l;
    } else if (n->name == "joint") {
      UrdfJoint j;
      j.name = n->get("name");
      j.type = n->get("type");
      if (const XmlNode* o = n->child("origin")) { parseTriple(o->get("xyz", "0 0 0"), j.xyz, "joint origin"); parseTriple(o->get("rpy", "0 0 0"), j.rpy, "joint rpy"); }
      if (const XmlNode* p = n->child("parent")) j.parent = p->get("link");
      if (const XmlNode* c = n->child("child")) j.child = c->get("link");
      if (const XmlNode* a = n->child("axis")) parseTriple(a->get("xyz", "1 0 0"), j.axis, "joint axis");
      if (const XmlNode* l = n->child("limit")) {
        j.lower = std::strtod(l->get("lower", "0").c_str(), nullptr);
        j.upper = std::strtod(l->get("upper", "0").c_str(), nullptr);
        j.effort = std::strtod(l->get("effort", "0").c_str(), nullptr);
        j.velocity = std::strtod(l->get("velocity", "0").c_str(), nullptr);
      }
      if (j.rpy[0] != 0 || j.rpy[1] != 0 || j.rpy[2] != 0) throw UnsupportedModel("joint " + j.name + " has a rotated origin (kernels assume rpy = 0)");
      joints[j.name] = j;
    }
  }
  if (links.empty()) throw std::runtime_error("URDF: no links in " + urdfFile);
  // root = link that is nobody's child
  std::set<std::string> childLinks;
  for (const auto& kv : joints) childLinks.insert(kv.second.child);
  std::string root;
  for (const auto& kv : links) if (!childLinks.count(kv.first)) { if (!root.empty()) throw UnsupportedModel("URDF has more than one root link"); root = kv.first; }
  if (root.empty()) throw UnsupportedModel("URDF has no root link");

  std::set<std::string> actuated(kJointNames, kJointNames + QMGPU_NJ);
  std::memset(&md, 0, sizeof(md));
  std::vector<Inertia> acc(QMGPU_NB);
  std::vector<std::string> bodyJoint(QMGPU_NB);
  std::map<std::string, std::pair<int, std::array<double, 3>>> frames;  // link name -> (body, offset in body frame)
  int numBodies = 1;
  md.parent[0] = -1;
  std::function<void(const std::string&, int, std::array<double, 3>)> visit = [&](const std::string& linkName, int body, std::array<double, 3> off) {
    const UrdfLink& l = links.at(linkName);
    frames[linkName] = {body, off};
    double I[3][3] = {{l.I[0], l.I[1], l.I[2]}, {l.I[1], l.I[3], l.I[4]}, {l.I[2], l.I[4], l.I[5]}};
    const double c[3] = {off[0] + l.com[0], off[1] + l.com[1], off[2] + l.com[2]};
    acc[body].add(l.mass, c, I);
    for (const auto& kv : joints) {  // alphabetical by joint name
      const UrdfJoint& j = kv.second;
      if (j.parent != linkName) continue;
      const std::array<double, 3> joff = {off[0] + j.xyz[0], off[1] + j.xyz[1], off[2] + j.xyz[2]};
      const bool moving = (j.type == "revolute" || j.type == "continuous") && actuated.count(j.name);
      if (!moving) {
        if (j.type != "fixed" && j.type != "revolute" && j.type != "continuous") throw UnsupportedModel("joint " + j.name + " has unsupported type " + j.type);
        visit(j.child, body, joff);  // fixed (or un-listed, frozen at q = 0): merge into the parent body
      } else {
        if (numBodies >= QMGPU_NB) throw UnsupportedModel("more than 18 actuated joints");
        const int nb = numBodies++;
        int ax = -1;
        for (int a = 0; a < 3; ++a) if (j.axis[a] == 1.0 && j.axis[(a + 1) % 3] == 0.0 && j.axis[(a + 2) % 3] == 0.0) ax = a;
        if (ax < 0) throw UnsupportedModel("joint " + j.name + " axis is not a positive unit coordinate axis");
        md.parent[nb] = body;
        md.axis[nb] = ax;
        for (int a = 0; a < 3; ++a) md.joint_offset[nb][a] = joff[a];
        bodyJoint[nb] = j.name;
        md.q_lower[nb - 1] = j.lower; md.q_upper[nb - 1] = j.upper; md.effort_limit[nb - 1] = j.effort; md.velocity_limit[nb - 1] = j.velocity;
        visit(j.child, nb, {0.0, 0.0, 0.0});
      }
    }
  };
  visit(root, 0, {0.0, 0.0, 0.0});
  if (numBodies != QMGPU_NB) throw UnsupportedModel("expected 18 actuated joints, found " + std::to_string(numBodies - 1));
  for (int b = 1; b < QMGPU_NB; ++b)
    if (bodyJoint[b] != kJointNames[b - 1]) throw UnsupportedModel("joint order mismatch at index " + std::to_string(b - 1) + ": got " + bodyJoint[b] + ", expected " + kJointNames[b - 1]);
  md.total_mass = 0.0;
  for (int b = 0; b < QMGPU_NB; ++b) {
    md.mass[b] = acc[b].m;
    md.total_mass += acc[b].m;
    for (int a = 0; a < 3; ++a) md.com[b][a] = acc[b].c[a];
    md.inertia[b][0] = acc[b].I[0][0]; md.inertia[b][1] = acc[b].I[0][1]; md.inertia[b][2] = acc[b].I[0][2];
    md.inertia[b][3] = acc[b].I[1][1]; md.inertia[b][4] = acc[b].I[1][2]; md.inertia[b][5] = acc[b].I[2][2];
  }
  for (int c = 0; c < QMGPU_NC; ++c) {
    auto it = frames.find(kContactNames[c]);
    if (it == frames.end()) throw UnsupportedModel(std::string("contact frame ") + kContactNames[c] + " not found");
    md.foot_body[c] = it->second.first;
    for (int a = 0; a < 3; ++a) md.foot_offset[c][a] = it->second.second[a];
  }
  auto it = frames.find(eeFrame);
  if (it == frames.end()) throw UnsupportedModel("end-effector frame '" + eeFrame + "' not found in the URDF");
  md.ee_body = it->second.first;
  for (int a = 0; a < 3; ++a) md.ee_offset[a] = it->second.second[a];
}

static void loadSettings(const std::string& taskFile, const std::string& referenceFile, const char* gainsFile, qmgpu_problem& P) {
  qmgpu_settings& s = P.settings;
  std::memset(&s, 0, sizeof(s));
  const InfoNode task = InfoParser::parseFile(taskFile);
  const InfoNode ref = InfoParser::parseFile(referenceFile);
  const int modelType = int(infoDoubleOr(task, "centroidalModelType", 0));
  if (modelType != 0) throw UnsupportedModel("only centroidalModelType 0 (full centroidal dynamics) is implemented");
  s.position_error_gain = infoDoubleOr(task, "model_settings.positionErrorGain", 0.0);              // ModelSettings.h:22
  s.phase_transition_stance_time = infoDoubleOr(task, "model_settings.phaseTransitionStanceTime", 0.4);
  s.liftoff_velocity = infoDouble(task, "swing_trajectory_config.liftOffVelocity");
  s.touchdown_velocity = infoDouble(task, "swing_trajectory_config.touchDownVelocity");
  s.swing_height = infoDouble(task, "swing_trajectory_config.swingHeight");
  s.touchdown_after_horizon = infoDoubleOr(task, "swing_trajectory_config.touchdownAfterHorizon", 0.2);
  s.swing_time_scale = infoDouble(task, "swing_trajectory_config.swingTimeScale");
  s.dt = infoDouble(task, "sqp.dt");
  s.sqp_iterations = int(infoDoubleOr(task, "sqp.sqpIteration", 1));
  s.delta_tol = infoDoubleOr(task, "sqp.deltaTol", 1e-6);
  s.cost_tol = infoDoubleOr(task, "sqp.costTol", 1e-4);
  s.g_max = infoDoubleOr(task, "sqp.g_max", 1e6);
  s.g_min = infoDoubleOr(task, "sqp.g_min", 1e-6);
  s.alpha_decay = 0.5; s.alpha_min = 1e-4; s.gamma_c = 1e-6; s.armijo_factor = 1e-4;  // upstream ocs2_sqp / FilterLinesearch defaults
  s.time_horizon = infoDoubleOr(task, "mpc.timeHorizon", 1.0);
  infoMatrix(task, "initialState", QMGPU_NX, 1, s.initial_state);
  infoMatrix(task, "Q", QMGPU_NX, QMGPU_NX, s.Q);
  infoMatrix(task, "R", QMGPU_NU, QMGPU_NU, s.R_task);
  s.ee_mu_position = infoDoubleOr(task, "endEffector.muPosition", 1.0);                               // QMInterface.cpp:149-156
  s.ee_mu_orientation = infoDoubleOr(task, "endEffector.muOrientation", 1.0);
  s.ee_final_mu_position = infoDoubleOr(task, "finalEndEffector.muPosition", 1.0);
  s.ee_final_mu_orientation = infoDoubleOr(task, "finalEndEffector.muOrientation", 1.0);
  s.friction_coefficient = infoDoubleOr(task, "frictionConeSoftConstraint.frictionCoefficient", 1.0);  // QMInterface.cpp:389-397
  s.friction_barrier_mu = infoDouble(task, "frictionConeSoftConstraint.mu");
  s.friction_barrier_delta = infoDouble(task, "frictionConeSoftConstraint.delta");
  s.friction_regularization = 25.0; s.friction_hessian_shift = 1e-6;  // upstream FrictionConeConstraint::Config defaults
  s.joint_pos_barrier_mu = infoDoubleOr(task, "jointPositionLimits.mu", 1e-2);                           // QMInterface.cpp:192-200
  s.joint_pos_barrier_delta = infoDoubleOr(task, "jointPositionLimits.delta", 1e-3);
  s.joint_vel_barrier_mu = infoDoubleOr(task, "jointVelocityLimits.mu", 1e-2);
  s.joint_vel_barrier_delta = infoDoubleOr(task, "jointVelocityLimits.delta", 1e-3);
  infoMatrix(task, "jointVelocityLimits.lowerBound.arm", 6, 1, s.arm_vel_lower);
  infoMatrix(task, "jointVelocityLimits.upperBound.arm", 6, 1, s.arm_vel_upper);
  s.wbc_friction_coefficient = infoDouble(task, "frictionConeTask.frictionCoefficient");                // WbcBase.cpp:617-622
  s.com_height = infoDouble(ref, "comHeight");
  infoMatrix(ref, "defaultJointState", QMGPU_NJ, 1, s.default_joint_state);
  s.target_displacement_velocity = infoDoubleOr(ref, "targetDisplacementVelocity", 0.5);
  s.target_rotation_velocity = infoDoubleOr(ref, "targetRotationVelocity", 0.3);
  s.gravity = 9.81;
  // force tracking (own formulation, include/qmgpu.h): absent in the reference's task.info -> off
  s.ee_contact_stiffness = infoDoubleOr(task, "forceTracking.stiffness", 0.0);
  s.ee_force_mu = infoDoubleOr(task, "forceTracking.muForce", 0.0);
  // ddp{} block (task.info:34-72): what the DDP variant of the solver uses of it
  s.ddp_min_step = infoDoubleOr(task, "ddp.lineSearch.minStepLength", 1e-2);
  s.ddp_max_step = infoDoubleOr(task, "ddp.lineSearch.maxStepLength", 1.0);
  s.ddp_constraint_penalty = infoDoubleOr(task, "ddp.constraintPenaltyInitialValue", 20.0);
  // WBC gains: defaults of qm_wbc/cfg/wbcWigeht.cfg:7-47, optionally overridden from an INFO block "wbc_gains"
  s.kp_swing = 350; s.kd_swing = 37; s.kp_base_height = 400; s.kd_base_height = 140; s.kp_base_linear = 400; s.kd_base_linear = 100;
  s.kp_base_angular = 400; s.kd_base_angular = 140;
  const double kpArm[6] = {4000, 4200, 4000, 4000, 4200, 6000};
  for (int i = 0; i < 6; ++i) { s.kp_arm_joint[i] = kpArm[i]; s.kd_arm_joint[i] = 75; }
  for (int i = 0; i < 3; ++i) { s.kp_ee_linear[i] = 3000; s.kd_ee_linear[i] = 75; s.kp_ee_angular[i] = 2000; s.kd_ee_angular[i] = 75; }
  if (gainsFile && gainsFile[0]) {
    const InfoNode g = InfoParser::parseFile(gainsFile);
    auto get = [&](const char* k, double& v) { v = infoDoubleOr(g, std::string("wbc_gains.") + k, v); };
    get("kp_swing", s.kp_swing); get("kd_swing", s.kd_swing); get("baseHeightKp", s.kp_base_height); get("baseHeightKd", s.kd_base_height);
    get("kp_base_linear", s.kp_base_linear); get("kd_base_linear", s.kd_base_linear); get("kp_base_angular", s.kp_base_angular); get("kd_base_angular", s.kd_base_angular);
    const char* ax[3] = {"x", "y", "z"};
    for (int i = 0; i < 6; ++i) { get(("kp_arm_joint_" + std::to_string(i + 1)).c_str(), s.kp_arm_joint[i]); get(("kd_arm_joint_" + std::to_string(i + 1)).c_str(), s.kd_arm_joint[i]); }
    for (int i = 0; i < 3; ++i) {
      get((std::string("kp_ee_linear_") + ax[i]).c_str(), s.kp_ee_linear[i]); get((std::string("kd_ee_linear_") + ax[i]).c_str(), s.kd_ee_linear[i]);
      get((std::string("kp_ee_angular_") + ax[i]).c_str(), s.kp_ee_angular[i]); get((std::string("kd_ee_angular_") + ax[i]).c_str(), s.kd_ee_angular[i]);
    }
  }
}

static int modeFromString(const std::string& s) {
  // upstream ocs2_legged_robot MotionPhaseDefinition.h string2ModeNumber
  static const std::map<std::string, int> m = {{"FLY", 0}, {"RH", 1}, {"LH", 2}, {"LH_RH", 3}, {"RF", 4}, {"RF_RH", 5}, {"RF_LH", 6}, {"RF_LH_RH", 7},
                                               {"LF", 8}, {"LF_RH", 9}, {"LF_LH", 10}, {"LF_LH_RH", 11}, {"LF_RF", 12}, {"LF_RF_RH", 13}, {"LF_RF_LH", 14}, {"STANCE", 15}};
  auto it = m.find(s);
  return it == m.end() ? -1 : it->second;
}

}  // namespace qmhost

using namespace qmhost;

extern "C" {

const char* qmgpu_strerror(int status) {
  switch (status) {
    case QMGPU_OK: return "ok";
    case QMGPU_ERR_INVALID_ARGUMENT: return "invalid argument";
    case QMGPU_ERR_FILE_NOT_FOUND: return "file not found";
    case QMGPU_ERR_PARSE: return "parse error";
    case QMGPU_ERR_UNSUPPORTED_MODEL: return "unsupported robot model";
    case QMGPU_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case QMGPU_ERR_HIP: return "HIP runtime error";
    case QMGPU_ERR_CAPACITY: return "capacity exceeded";
    case QMGPU_ERR_NUMERICAL: return "numerical failure";
    default: return "unknown status";
  }
}

const char* qmgpu_last_error(void) { return g_lastError.c_str(); }

int qmgpu_load_problem(const char* task_file, const char* urdf_file, const char* reference_file, const char* wbc_gains_file, qmgpu_problem* out) {
  if (!task_file || !urdf_file || !reference_file || !out) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null argument");
  return guarded([&]() {
    // existence checks first, in the reference's order (QMInterface.cpp:40-62)
    readFile(task_file); readFile(urdf_file); readFile(reference_file);
    loadSettings(task_file, reference_file, wbc_gains_file, *out);
    // end-effector frame: model_settings.eeFrame (ModelSettings.cpp:31), used at QMInterface.cpp:163-165
    std::string eeFrame = "z1_end_effector";
    const InfoNode task = InfoParser::parseFile(task_file);
    if (const InfoNode* n = task.find("model_settings.eeFrame")) if (!n->value.empty()) eeFrame = n->value;
    buildModel(urdf_file, eeFrame, out->model);
  });
}

int qmgpu_mode_from_string(const char* name) { return name ? modeFromString(name) : -1; }

int qmgpu_load_gait(const char* gait_file, const char* gait_name, qmgpu_gait* out) {
  if (!gait_file || !gait_name || !out) return setError(QMGPU_ERR_INVALID_ARGUMENT, "null argument");
  return guarded([&]() {
    const InfoNode root = InfoParser::parseFile(gait_file);
    const InfoNode* g = root.child(gait_name);
    if (!g) throw std::runtime_error(std::string("gait '") + gait_name + "' not found in " + gait_file);
    const InfoNode* seq = g->child("modeSequence");
    const InfoNode* sw = g->child("switchingTimes");
    if (!seq || !sw) throw std::runtime_error("gait needs modeSequence and switchingTimes");
    std::memset(out, 0, sizeof(*out));
    const int n = int(seq->children.size());
    if (n < 1 || n > QMGPU_MAX_EVENTS || int(sw->children.size()) != n + 1) throw std::runtime_error("gait template has inconsistent sizes");
    out->num_modes = n;
    for (int i = 0; i < n; ++i) {
      const int m = modeFromString(seq->children[i].value);
      if (m < 0) throw std::runtime_error("unknown mode name " + seq->children[i].value);
      out->modes[i] = m;
    }
    for (int i = 0; i <= n; ++i) out->switching_times[i] = std::strtod(sw->children[i].value.c_str(), nullptr);
  });
}

// Mode schedule over [t_begin, t_end]: STANCE until t_phase0, then the template tiled until it covers t_end, then the
// default final STANCE phase -- the shape upstream GaitSchedule::tileModeSequenceTemplate produces after the initial
// STANCE schedule of reference.info:24-36 (loaded at QMInterface.cpp:455-480).
int qmgpu_tile_gait(const qmgpu_gait* gait, double t_phase0, double t_begin, double t_end, int32_t* num_events, double* event_times, int32_t* modes) {
  return qmgpu_switch_gait(gait, 15, 0.0, t_phase0, t_begin, t_end, num_events, event_times, modes);
}

// A gait command arriving while another mode is running (GaitTopicPublisher.cpp:31-44 -> upstream GaitReceiver -> GaitSchedule::insertModeSequenceTemplate):
// the running mode `prev_mode` lasts until t_switch; unless it is STANCE or equals the template's first mode, a STANCE phase of
// `transition_stance_time` (model_settings.phaseTransitionStanceTime, task.info:11) is inserted; then the template is tiled.
int qmgpu_switch_gait(const qmgpu_gait* gait, int32_t prev_mode, double transition_stance_time, double t_switch, double t_begin, double t_end, int32_t* num_events,
                      double* event_times, int32_t* modes) {
  if (!gait || !num_events || !event_times || !modes || gait->num_modes < 1 || gait->num_modes > QMGPU_MAX_EVENTS || prev_mode < 0 || prev_mode > 15)
    return setError(QMGPU_ERR_INVALID_ARGUMENT, "bad gait (1 <= num_modes <= QMGPU_MAX_EVENTS, 0 <= prev_mode <= 15)");
  if (!std::isfinite(t_switch) || !std::isfinite(t_begin) || !std::isfinite(t_end) || !std::isfinite(transition_stance_time)) return setError(QMGPU_ERR_INVALID_ARGUMENT, "gait times must be finite");
  const bool transition = prev_mode != 15 && prev_mode != gait->modes[0] && transition_stance_time > 0.0;
  const double t_phase0 = transition ? t_switch + transition_stance_time : t_switch;
  const double period = gait->switching_times[gait->num_modes] - gait->switching_times[0];
  if (!(period > 0.0) || !std::isfinite(period)) return setError(QMGPU_ERR_INVALID_ARGUMENT, "gait period must be positive and finite");
  std::vector<double> ev;
  std::vector<int> md;
  // first tiled cycle: the one BEFORE the cycle that contains t_begin (so that a swing phase straddling the template boundary keeps
  // its real lift-off time, as upstream GaitSchedule keeps the preceding modes), but never before t_phase0: STANCE only precedes
  // t_phase0 itself
  double start = t_phase0;
  if (t_begin > t_phase0) {
    const double cycles = std::floor((t_begin - t_phase0) / period);
    start = t_phase0 + (cycles >= 1.0 ? cycles - 1.0 : 0.0) * period;
  }
  md.push_back(prev_mode);   // the mode running before the switch (STANCE for a plain tiling)
  if (transition && start == t_phase0) { ev.push_back(t_switch); md.push_back(15); }   // the inserted stance phase (only while the horizon still sees the switch)
  else if (start > t_phase0) md.back() = 15;   // the horizon starts cycles after the switch: what precedes the first tiled cycle is never looked up
  ev.push_back(start);
  double t = start;
  // bounded like the device tiler (frontend_kernel.h): more than QMGPU_MAX_EVENTS + 2 cycles either overflow the event table or add nothing
  bool truncated = false;
  for (int cycle = 0; t < t_end; ++cycle) {
    if (cycle >= QMGPU_MAX_EVENTS + 2) { truncated = true; break; }
    for (int i = 0; i < gait->num_modes; ++i) {
      md.push_back(gait->modes[i]);
      t += gait->switching_times[i + 1] - gait->switching_times[i];
      ev.push_back(t);
    }
  }
  // a truncated tiling that did not overflow is a template without mode changes (every cycle merged away): the default final phase still starts at
  // the end of the last WHOLE cycle at or after t_end, not where the bounded loop happened to stop (a one-mode non-stance template with a short period)
  if (truncated && t < t_end) { t += std::ceil((t_end - t) / period) * period; ev.back() = t; }
  md.push_back(15);  // default final phase
  // merge equal neighbouring modes (e.g. a pure stance template)
  std::vector<double> ev2;
  std::vector<int> md2{md[0]};
  for (size_t i = 0; i < ev.size(); ++i) {
    if (md[i + 1] == md2.back()) continue;
    ev2.push_back(ev[i]);
    md2.push_back(md[i + 1]);
  }
  if (int(ev2.size()) > QMGPU_MAX_EVENTS) return setError(QMGPU_ERR_CAPACITY, "mode schedule needs more than QMGPU_MAX_EVENTS events");
  *num_events = int(ev2.size());
  for (size_t i = 0; i < ev2.size(); ++i) event_times[i] = ev2[i];
  for (size_t i = 0; i < md2.size(); ++i) modes[i] = md2[i];
  for (size_t i = md2.size(); i < size_t(QMGPU_MAX_EVENTS) + 1; ++i) modes[i] = 15;
  for (size_t i = ev2.size(); i < size_t(QMGPU_MAX_EVENTS); ++i) event_times[i] = 1e300;
  return QMGPU_OK;
}

int qmgpu_time_grid_with_events(double t0, double tf, double dt, int32_t num_events, const double* event_times, int32_t max_nodes, int32_t* n_out, double* grid) {
  if (!(dt > 0.0) || !(tf > t0) || !n_out || !grid || (num_events > 0 && !event_times)) return setError(QMGPU_ERR_INVALID_ARGUMENT, "bad time grid arguments");
  const double eps = 1e-3 * dt;
  std::vector<double> g{t0};
  int nextEv = 0;
  while (nextEv < num_events && event_times[nextEv] <= t0 + eps) ++nextEv;   // events at or before t0 are already in force
  while (g.back() < tf - eps) {
    double t = g.back() + dt;
    if (nextEv < num_events && event_times[nextEv] <= t + eps && event_times[nextEv] < tf - eps) t = event_times[nextEv++];
    if (t > tf - eps) t = tf;
    g.push_back(t);
  }
  if (int(g.size()) - 1 > max_nodes) return setError(QMGPU_ERR_CAPACITY, "time grid needs more nodes than max_nodes");
  *n_out = int(g.size()) - 1;
  for (size_t i = 0; i < g.size(); ++i) grid[i] = g[i];
  return QMGPU_OK;
}

}  // extern "C"
