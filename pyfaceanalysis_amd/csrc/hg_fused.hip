#include "hg_common.hpp"
namespace hg {
std::unique_ptr<Executor> make_fused_executor(const TNode&, std::string* why_not) {
    if (why_not) *why_not = "fused plan not built yet";
    return nullptr;
}
}  // namespace hg
