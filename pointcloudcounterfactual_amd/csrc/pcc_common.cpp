// Per-thread error record and library info for libpcc_structural.so.
#include "pcc_common.hpp"

namespace {
thread_local int t_status = 0;
thread_local char t_msg[320] = "";
}  // namespace

namespace pcc {
void set_error(int status, const char *what) {
    t_status = status;
    std::strncpy(t_msg, what ? what : "", sizeof t_msg - 1);
    t_msg[sizeof t_msg - 1] = '\0';
}
void clear_error() {
    t_status = 0;
    t_msg[0] = '\0';
}
}  // namespace pcc

extern "C" {
const char *pcc_version(void) { return "pcc_structural 0.1 (gfx950)"; }
const char *pcc_last_error(void) { return t_msg; }
int pcc_last_status(void) { return t_status; }
}
