// pdlp_env.hpp — the master switch of the development environment variables.
#pragma once

namespace pdlp {

// Development and test switches (layouts, variants kept for A/B measurements, fault injection, profiles) are looked at only
// when PDLP_MI355X_DEV is set to a non-zero value: a stray variable in a user's environment cannot change the path a solve
// takes.  A development variable that is set without the master switch is reported once on stderr and ignored.  The
// switches a user may need are read with getenv directly.
const char* devEnv(const char* name);

}  // namespace pdlp
