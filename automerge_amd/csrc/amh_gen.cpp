// placeholder, replaced below
#include "../../include/automerge_hip.h"
extern "C" int amh_gen_map_docs(uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, int, amh_bytes_out*, uint64_t**) { return AMH_ERR_UNIMPLEMENTED; }
