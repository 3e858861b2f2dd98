// TEST INFRASTRUCTURE ONLY -- stand-in of upstream ocs2_oc/oc_problem/OptimalControlProblem.h: a copyable value (upstream deep-copies its cost /
// constraint collections); the adapters only store and hand it back (SolverBase::getOptimalControlProblem).
#pragma once
namespace ocs2 {
struct OptimalControlProblem { int id = 0; };
}  // namespace ocs2
