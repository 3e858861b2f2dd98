#pragma once
#include "ocs2_core/reference/ModeSchedule.h"
namespace ocs2 { namespace legged_robot {
class GaitSchedule {};   // upstream ocs2_legged_robot/gait/GaitSchedule.h (opaque to the adapters)
}}
