// SGD kernel instantiations for row-group shape VEC=1, G=64, KPL=2 (see rfm_sgd.hpp)
#define RFM_VEC 1
#define RFM_G 64
#define RFM_KPL 2
#define RFM_SHAPE_FN sgd_table_v1_g64_k2
#include "rfm_sgd_inst.inc"
