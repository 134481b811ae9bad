// placeholder until the MFMA tower lands
#include "az_net.h"
#include <string>
namespace az {
int fail(int code, const std::string& msg);
bool mfma_supported(const az_net_config&) { return false; }
int mfma_create(const az_net_config&, const NetLayout&, const float*, MfmaNet**) { return fail(AZ_ERR_ARG, "bf16 path not built"); }
void mfma_destroy(MfmaNet*) {}
int mfma_forward(MfmaNet*, const void*, int, float*, float*, hipStream_t) { return fail(AZ_ERR_ARG, "bf16 path not built"); }
}
