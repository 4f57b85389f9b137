// pcd_producer.hip - device operator producer (pcd_fe_*)
// (one of the engine's translation units; shared declarations: pcd_internal.hpp)
#include "pcd_internal.hpp"

#include "pcd_producer_abi.hpp"
