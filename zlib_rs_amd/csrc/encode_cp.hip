// encode_cp.hip -- the encoder's cost-parse instantiation (levels 3-9) as a translation unit of its own: see the note at
// zmi_launch_encode_cp in encode.hip (one source, two units; every helper is `static`, so each unit has its own copies).
#define ENC_CP_UNIT
#include "encode.hip"
