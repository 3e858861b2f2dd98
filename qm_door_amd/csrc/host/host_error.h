// Error plumbing between the C++ host code and the C ABI: exceptions never cross extern "C".
// Mirrors the reference's error behaviour (std::invalid_argument for missing files, QMInterface.cpp:45,53,61;
// std::runtime_error elsewhere) as status codes + a thread-local message.
#pragma once
#include <stdexcept>
#include <string>

#include "../../../include/qmgpu.h"

namespace qmhost {

extern thread_local std::string g_lastError;

struct UnsupportedModel : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct HipFailure : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct NoDevice : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct CapacityError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

inline int setError(int status, const std::string& msg) {
  g_lastError = msg;
  return status;
}

template <class F>
int guarded(F&& f) {
  try {
    f();
    return QMGPU_OK;
  } catch (const std::invalid_argument& e) {
    const std::string m = e.what();
    return setError(m.rfind("file not found", 0) == 0 ? QMGPU_ERR_FILE_NOT_FOUND : QMGPU_ERR_INVALID_ARGUMENT, m);
  } catch (const UnsupportedModel& e) {
    return setError(QMGPU_ERR_UNSUPPORTED_MODEL, e.what());
  } catch (const NoDevice& e) {
    return setError(QMGPU_ERR_NO_DEVICE, e.what());
  } catch (const HipFailure& e) {
    return setError(QMGPU_ERR_HIP, e.what());
  } catch (const CapacityError& e) {
    return setError(QMGPU_ERR_CAPACITY, e.what());
  } catch (const std::exception& e) {
    return setError(QMGPU_ERR_PARSE, e.what());
  } catch (...) {
    return setError(QMGPU_ERR_PARSE, "unknown exception");
  }
}

}  // namespace qmhost
