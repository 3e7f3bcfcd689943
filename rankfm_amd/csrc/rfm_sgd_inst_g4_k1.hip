// SGD kernel instantiations for row-group shape G=4 lanes x KPL=1 dwords per lane (see rfm_sgd.hpp)
#define RFM_G 4
#define RFM_KPL 1
#define RFM_SHAPE_FN sgd_table_g4_k1
#include "rfm_sgd_inst.inc"
