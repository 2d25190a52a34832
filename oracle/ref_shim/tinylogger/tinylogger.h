// oracle/ref_shim (TEST INFRASTRUCTURE ONLY): dependencies/tinylogger is an empty submodule in the reference mount; the compiled sources only stream status lines into it
#pragma once
namespace tlog {
struct Sink { template <typename T> Sink& operator<<(const T&) { return *this; } };
inline Sink info() { return {}; }
inline Sink success() { return {}; }
inline Sink warning() { return {}; }
inline Sink error() { return {}; }
inline Sink debug() { return {}; }
inline Sink none() { return {}; }
} // namespace tlog
