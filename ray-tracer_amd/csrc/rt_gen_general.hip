// Instantiations of the generation kernel (rt_generation.h): one translation unit per group, so that
// they compile side by side.
#include "rt_generation.h"

namespace rt {
template void launch_generation_variant<false, true, false, false>(const SceneDev&, unsigned, int, const GenLaunch&);
}  // namespace rt
