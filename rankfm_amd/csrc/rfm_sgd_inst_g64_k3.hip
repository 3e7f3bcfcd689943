// SGD kernel instantiations for row-group shape G=64 lanes x KPL=3 dwords per lane (see rfm_sgd.hpp)
#define RFM_G 64
#define RFM_KPL 3
#define RFM_SHAPE_FN sgd_table_g64_k3
#include "rfm_sgd_inst.inc"
