#include "cbfssm_adjoint_inst.hpp"
CBF_REV_INSTANTIATE(20)
