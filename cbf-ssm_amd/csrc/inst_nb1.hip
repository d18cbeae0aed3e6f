#include "cbfssm_inst.hpp"
CBF_INSTANTIATE(1)
