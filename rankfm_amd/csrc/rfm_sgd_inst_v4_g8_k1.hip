// SGD kernel instantiations for row-group shape VEC=4, G=8, KPL=1 (see rfm_sgd.hpp)
#define RFM_VEC 4
#define RFM_G 8
#define RFM_KPL 1
#define RFM_SHAPE_FN sgd_table_v4_g8_k1
#include "rfm_sgd_inst.inc"
