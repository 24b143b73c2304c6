// hip/solver/{fcg,bicgstab,cgs,bicg,ir}_kernels.hip.cpp (the HIP instantiations of
// common/unified/solver/{fcg,bicgstab,cgs,bicg,ir}_kernels.cpp): the step kernels of the remaining Krylov
// solvers (core/solver/{fcg,bicgstab,cgs,bicg,ir}_kernels.hpp).  One file here, one per solver in a reference tree.
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace {
inline uint8_t* raw(array<stopping_status>* s) { return reinterpret_cast<uint8_t*>(s->get_data()); }
inline const uint8_t* raw(const array<stopping_status>* s) { return reinterpret_cast<const uint8_t*>(s->get_const_data()); }
using Vec = matrix::Dense<double>;
#define V(m) (m)->get_values(), (m)->get_stride()
#define C(m) (m)->get_const_values(), (m)->get_stride()
}  // namespace

namespace fcg {

void initialize(std::shared_ptr<const HipExecutor> exec, const Vec* b, Vec* r, Vec* z, Vec* p, Vec* q, Vec* t, Vec* prev_rho, Vec* rho,
                Vec* rho_t, array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_fcg_initialize_f64(GKOMI_NULL_STREAM, b->get_size()[0], b->get_size()[1], C(b), V(r), V(z), V(p), V(q), V(t),
                                        prev_rho->get_values(), rho->get_values(), rho_t->get_values(), raw(stop_status)));
}

void step_1(std::shared_ptr<const HipExecutor> exec, Vec* p, const Vec* z, const Vec* rho_t, const Vec* prev_rho,
            const array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_fcg_step_1_f64(GKOMI_NULL_STREAM, p->get_size()[0], p->get_size()[1], V(p), C(z), rho_t->get_const_values(),
                                    prev_rho->get_const_values(), raw(stop_status)));
}

void step_2(std::shared_ptr<const HipExecutor> exec, Vec* x, Vec* r, Vec* t, const Vec* p, const Vec* q, const Vec* beta, const Vec* rho,
            const array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_fcg_step_2_f64(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], V(x), V(r), V(t), C(p), C(q),
                                    beta->get_const_values(), rho->get_const_values(), raw(stop_status)));
}

}  // namespace fcg

namespace bicgstab {

void initialize(std::shared_ptr<const HipExecutor> exec, const Vec* b, Vec* r, Vec* rr, Vec* y, Vec* s, Vec* t, Vec* z, Vec* v, Vec* p,
                Vec* prev_rho, Vec* rho, Vec* alpha, Vec* beta, Vec* gamma, Vec* omega, array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_bicgstab_initialize_f64(GKOMI_NULL_STREAM, b->get_size()[0], b->get_size()[1], C(b), V(r), V(rr), V(y), V(s), V(t),
                                             V(z), V(v), V(p), prev_rho->get_values(), rho->get_values(), alpha->get_values(),
                                             beta->get_values(), gamma->get_values(), omega->get_values(), raw(stop_status)));
}

void step_1(std::shared_ptr<const HipExecutor> exec, const Vec* r, Vec* p, const Vec* v, const Vec* rho, const Vec* prev_rho,
            const Vec* alpha, const Vec* omega, const array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_bicgstab_step_1_f64(GKOMI_NULL_STREAM, r->get_size()[0], r->get_size()[1], C(r), V(p), C(v), rho->get_const_values(),
                                         prev_rho->get_const_values(), alpha->get_const_values(), omega->get_const_values(),
                                         raw(stop_status)));
}

void step_2(std::shared_ptr<const HipExecutor> exec, const Vec* r, Vec* s, const Vec* v, const Vec* rho, Vec* alpha, const Vec* beta,
            const array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_bicgstab_step_2_f64(GKOMI_NULL_STREAM, r->get_size()[0], r->get_size()[1], C(r), V(s), C(v), rho->get_const_values(),
                                         alpha->get_values(), beta->get_const_values(), raw(stop_status)));
}

void step_3(std::shared_ptr<const HipExecutor> exec, Vec* x, Vec* r, const Vec* s, const Vec* t, const Vec* y, const Vec* z,
            const Vec* alpha, const Vec* beta, const Vec* gamma, Vec* omega, const array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_bicgstab_step_3_f64(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], V(x), V(r), C(s), C(t), C(y), C(z),
                                         alpha->get_const_values(), beta->get_const_values(), gamma->get_const_values(),
                                         omega->get_values(), raw(stop_status)));
}

void finalize(std::shared_ptr<const HipExecutor> exec, Vec* x, const Vec* y, const Vec* alpha, array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_bicgstab_finalize_f64(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], V(x), C(y), alpha->get_const_values(),
                                           raw(stop_status)));
}

}  // namespace bicgstab

namespace cgs {

void initialize(std::shared_ptr<const HipExecutor> exec, const Vec* b, Vec* r, Vec* r_tld, Vec* p, Vec* q, Vec* u, Vec* u_hat, Vec* v_hat,
                Vec* t, Vec* alpha, Vec* beta, Vec* gamma, Vec* prev_rho, Vec* rho, array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_cgs_initialize_f64(GKOMI_NULL_STREAM, b->get_size()[0], b->get_size()[1], C(b), V(r), V(r_tld), V(p), V(q), V(u),
                                        V(u_hat), V(v_hat), V(t), alpha->get_values(), beta->get_values(), gamma->get_values(),
                                        prev_rho->get_values(), rho->get_values(), raw(stop_status)));
}

void step_1(std::shared_ptr<const HipExecutor> exec, const Vec* r, Vec* u, Vec* p, const Vec* q, Vec* beta, const Vec* rho,
            const Vec* rho_prev, const array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_cgs_step_1_f64(GKOMI_NULL_STREAM, r->get_size()[0], r->get_size()[1], C(r), V(u), V(p), C(q), beta->get_values(),
                                    rho->get_const_values(), rho_prev->get_const_values(), raw(stop_status)));
}

void step_2(std::shared_ptr<const HipExecutor> exec, const Vec* u, const Vec* v_hat, Vec* q, Vec* t, Vec* alpha, const Vec* rho,
            const Vec* gamma, const array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_cgs_step_2_f64(GKOMI_NULL_STREAM, u->get_size()[0], u->get_size()[1], C(u), C(v_hat), V(q), V(t), alpha->get_values(),
                                    rho->get_const_values(), gamma->get_const_values(), raw(stop_status)));
}

void step_3(std::shared_ptr<const HipExecutor> exec, const Vec* t, const Vec* u_hat, Vec* r, Vec* x, const Vec* alpha,
            const array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_cgs_step_3_f64(GKOMI_NULL_STREAM, t->get_size()[0], t->get_size()[1], C(t), C(u_hat), V(r), V(x),
                                    alpha->get_const_values(), raw(stop_status)));
}

}  // namespace cgs

namespace bicg {

void initialize(std::shared_ptr<const HipExecutor> exec, const Vec* b, Vec* r, Vec* z, Vec* p, Vec* q, Vec* prev_rho, Vec* rho, Vec* r2,
                Vec* z2, Vec* p2, Vec* q2, array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_bicg_initialize_f64(GKOMI_NULL_STREAM, b->get_size()[0], b->get_size()[1], C(b), V(r), V(z), V(p), V(q),
                                         prev_rho->get_values(), rho->get_values(), V(r2), V(z2), V(p2), V(q2), raw(stop_status)));
}

void step_1(std::shared_ptr<const HipExecutor> exec, Vec* p, const Vec* z, Vec* p2, const Vec* z2, const Vec* rho, const Vec* prev_rho,
            const array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_bicg_step_1_f64(GKOMI_NULL_STREAM, p->get_size()[0], p->get_size()[1], V(p), C(z), V(p2), C(z2),
                                     rho->get_const_values(), prev_rho->get_const_values(), raw(stop_status)));
}

void step_2(std::shared_ptr<const HipExecutor> exec, Vec* x, Vec* r, Vec* r2, const Vec* p, const Vec* q, const Vec* q2, const Vec* beta,
            const Vec* rho, const array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_bicg_step_2_f64(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], V(x), V(r), V(r2), C(p), C(q), C(q2),
                                     beta->get_const_values(), rho->get_const_values(), raw(stop_status)));
}

}  // namespace bicg

namespace ir {

void initialize(std::shared_ptr<const HipExecutor> exec, array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_ir_initialize(GKOMI_NULL_STREAM, static_cast<int64_t>(stop_status->get_num_elems()), raw(stop_status)));
}

}  // namespace ir

namespace set_all_statuses {

// core/stop/criterion_kernels.hpp: stop::Iteration and stop::Combined end through it
void set_all_statuses(std::shared_ptr<const HipExecutor> exec, uint8 stoppingId, bool setFinalized, array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_set_all_statuses(GKOMI_NULL_STREAM, static_cast<int64_t>(stop_status->get_num_elems()), stoppingId,
                                      setFinalized ? 1 : 0, raw(stop_status)));
}

}  // namespace set_all_statuses
#undef V
#undef C
}  // namespace hip
}  // namespace kernels
}  // namespace gko
