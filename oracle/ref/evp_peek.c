/* TEST INFRASTRUCTURE (oracle/ref): read-only access to module-private allocatable arrays of the reference's
 * ice_dyn_evp (cicecore/cicedyn/dynamics/ice_dyn_evp.F90:59-112) for the fixture generator.  On the C grid the
 * inputs of the subcycle loop (ocean currents, drag, mass and forcing at E and N points ...) never cross a module
 * boundary -- there is no dyn_evp1d_run-style call to capture them at -- but they are ordinary data symbols of the
 * object file (flang: _QM<module>E<name>, an ISO_Fortran_binding descriptor whose first member is the base
 * address).  The harness dumps them after a preparation-only evp() call (ndte = 0).  Nothing of the reference is
 * copied or modified. */
#include <stddef.h>
#define DESC(n) extern char _QMice_dyn_evpE##n[];
DESC(uocne) DESC(vocne) DESC(cdn_ocne) DESC(waterxe) DESC(forcexe) DESC(aie) DESC(rheofacte) DESC(emassdti)
DESC(uocnn) DESC(vocnn) DESC(cdn_ocnn) DESC(wateryn) DESC(forceyn) DESC(ain) DESC(rheofactn) DESC(nmassdti)
DESC(zetax2t) DESC(etax2t) DESC(etax2u) DESC(shearu) DESC(deltau)
#define BASE(n) (*(void **)_QMice_dyn_evpE##n)
void *evp_peek_base(int which)
{
    switch (which) {
    case 1: return BASE(uocne);     case 2: return BASE(vocne);    case 3: return BASE(cdn_ocne);
    case 4: return BASE(waterxe);   case 5: return BASE(forcexe);  case 6: return BASE(aie);
    case 7: return BASE(rheofacte); case 8: return BASE(emassdti);
    case 9: return BASE(uocnn);     case 10: return BASE(vocnn);   case 11: return BASE(cdn_ocnn);
    case 12: return BASE(wateryn);  case 13: return BASE(forceyn); case 14: return BASE(ain);
    case 15: return BASE(rheofactn); case 16: return BASE(nmassdti);
    case 17: return BASE(zetax2t);  case 18: return BASE(etax2t);  case 19: return BASE(etax2u);
    case 20: return BASE(shearu);   case 21: return BASE(deltau);
    default: return NULL;
    }
}
