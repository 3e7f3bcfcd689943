// SGD kernel instantiations for row-group shape G=32 lanes x KPL=2 dwords per lane (see rfm_sgd.hpp)
#define RFM_VEC 1
#define RFM_G 32
#define RFM_KPL 2
#define RFM_SHAPE_FN sgd_table_g32_k2
#include "rfm_sgd_inst.inc"
