/* Umbrella header of the CSiftmi system-library target: the C ABI itself lives in include/siftmi.h of this repository
   (install it next to this file, or add its directory to the header search path: `swift build -Xcc -I<repo>/include`). */
#include "../../../include/siftmi.h"
