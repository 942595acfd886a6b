// C exports of the reference's own `*_kernel_wrapper` functions, compiled for the CPU from the
// reference sources (see build_ref.py).  The declarations below repeat the reference's
// (they live in its .cpp files: sampling.cpp:4-14, ball_query.cpp:4-6, group_points.cpp:4-10,
// interpolate.cpp:4-12).  TEST INFRASTRUCTURE ONLY.
void gather_points_kernel_wrapper(int b, int c, int n, int npoints, const float *points, const int *idx,
                                  float *out);
void gather_points_grad_kernel_wrapper(int b, int c, int n, int npoints, const float *grad_out,
                                       const int *idx, float *grad_points);
void furthest_point_sampling_kernel_wrapper(int b, int n, int m, const float *dataset, float *temp,
                                            int *idxs);
void query_ball_point_kernel_wrapper(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                     const float *xyz, int *idx);
void group_points_kernel_wrapper(int b, int c, int n, int npoints, int nsample, const float *points,
                                 const int *idx, float *out);
void group_points_grad_kernel_wrapper(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                      const int *idx, float *grad_points);
void three_nn_kernel_wrapper(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                             int *idx);
void three_interpolate_kernel_wrapper(int b, int c, int m, int n, const float *points, const int *idx,
                                      const float *weight, float *out);
void three_interpolate_grad_kernel_wrapper(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                           const float *weight, float *grad_points);

extern "C" {
void ref_gather_points(int b, int c, int n, int m, const float *p, const int *i, float *o) {
  gather_points_kernel_wrapper(b, c, n, m, p, i, o);
}
void ref_gather_points_grad(int b, int c, int n, int m, const float *g, const int *i, float *o) {
  gather_points_grad_kernel_wrapper(b, c, n, m, g, i, o);
}
void ref_furthest_point_sampling(int b, int n, int m, const float *d, float *t, int *i) {
  furthest_point_sampling_kernel_wrapper(b, n, m, d, t, i);
}
void ref_query_ball_point(int b, int n, int m, float r, int ns, const float *nx, const float *x, int *i) {
  query_ball_point_kernel_wrapper(b, n, m, r, ns, nx, x, i);
}
void ref_group_points(int b, int c, int n, int np, int ns, const float *p, const int *i, float *o) {
  group_points_kernel_wrapper(b, c, n, np, ns, p, i, o);
}
void ref_group_points_grad(int b, int c, int n, int np, int ns, const float *g, const int *i, float *o) {
  group_points_grad_kernel_wrapper(b, c, n, np, ns, g, i, o);
}
void ref_three_nn(int b, int n, int m, const float *u, const float *k, float *d, int *i) {
  three_nn_kernel_wrapper(b, n, m, u, k, d, i);
}
void ref_three_interpolate(int b, int c, int m, int n, const float *p, const int *i, const float *w, float *o) {
  three_interpolate_kernel_wrapper(b, c, m, n, p, i, w, o);
}
void ref_three_interpolate_grad(int b, int c, int n, int m, const float *g, const int *i, const float *w,
                                float *o) {
  three_interpolate_grad_kernel_wrapper(b, c, n, m, g, i, w, o);
}
}
