! Stand-in for MOM6's config_src/memory/dynamic_symmetric/MOM_memory.h: the array-extent macros the shims' dummy
! argument declarations use (dynamic, symmetric memory).
#define SYMMETRIC_MEMORY_
#define SZI_(G)  G%isd:G%ied
#define SZJ_(G)  G%jsd:G%jed
#define SZK_(G)  G%ke
#define SZIB_(G) G%IsdB:G%IedB
#define SZJB_(G) G%JsdB:G%JedB
