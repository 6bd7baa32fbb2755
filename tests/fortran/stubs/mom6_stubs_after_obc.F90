!> The stand-ins of mom6_stubs.F90 that import MOM_open_boundary, for builds with the reference's own MOM_open_boundary.F90 (-DREF_OBC):
!! compiled after it.  Nothing of this is used by the library or its shims.
#include <MOM_memory.h>
#include "mom6_stubs_after_obc.inc"
