// the builder drivers link nt_builder.cpp alone: nt_last_error() lives in nt_api.cpp
extern "C" const char *nt_last_error(void) { return ""; }
