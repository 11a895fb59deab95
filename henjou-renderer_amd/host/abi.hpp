// Sized structs of the C-ABI (include/henjou_hip.h, "Sized structs"): every hjr_scene_view / hjr_render_option / hjr_params /
// hjr_stats that crosses the boundary is copied through these two functions, so the library never reads or writes more bytes
// than the caller's struct_size says it owns.  A caller built against an older (shorter) header keeps working against a newer
// library and the other way round; before this rule an `hjr_stats` on the stack of a stale tools/kbench was overrun by the
// fields round 2 had appended (profiles/r03_experiments.md).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>

namespace hjr {
void set_error(const std::string& s);

inline bool abi_size(const void* user, uint32_t& n, const char* who)
{
    if (!user) { set_error(std::string(who) + ": null struct"); return false; }
    memcpy(&n, user, sizeof(n));
    if (n < 8u) { set_error(std::string(who) + ": struct_size is not set (initialise the struct with HJR_INIT)"); return false; }
    return true;
}
// input: local = the caller's fields, zero (= default) for those its struct does not have
template <class T> inline bool abi_take(const T* user, T& local, const char* who)
{
    uint32_t n;
    if (!abi_size(user, n, who)) return false;
    memset(static_cast<void*>(&local), 0, sizeof(T));
    memcpy(static_cast<void*>(&local), user, std::min<size_t>(n, sizeof(T)));
    local.struct_size = (uint32_t)sizeof(T);
    return true;
}
// output: the first min(caller's struct_size, sizeof(T)) bytes of `local`; the caller's struct_size stays what it was
template <class T> inline bool abi_give(T* user, const T& local, const char* who)
{
    uint32_t n;
    if (!abi_size(user, n, who)) return false;
    T tmp = local;
    tmp.struct_size = n;
    memcpy(static_cast<void*>(user), &tmp, std::min<size_t>(n, sizeof(T)));
    return true;
}
} // namespace hjr
