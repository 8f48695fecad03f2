// stubs.hip — entry points declared in include/selfocc_hip.h whose kernels are not built yet.
// They fail loudly (rc = -2); each is removed from here when its kernel lands.
#include "so_device.h"
#define SO_STUB(name) so_set_error(#name " is not implemented in this build"); return -2;
extern "C" int selfocc_reproj_fwd(const so_reproj_args *, void *) { SO_STUB(selfocc_reproj_fwd) }
extern "C" int selfocc_reproj_bwd(const so_reproj_args *, const float *, const float *, float *, void *) { SO_STUB(selfocc_reproj_bwd) }
