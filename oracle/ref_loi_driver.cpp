// ref_loi_driver.cpp -- TEST INFRASTRUCTURE ONLY (this container only).
//
// Links against the reference's own LOI.cpp, compiled from where it lies under
// /root/reference by oracle/Makefile (outputs only in oracle/_ref/), and calls
// its reorder functions on a CSR graph read from a small binary file.  Used by
// tests/golden/make_loi_fixtures.py to emit the permutation fixtures that pin
// oracle/loi_oracle.py and the product's hcspmm_loi_reorder.  The reference
// tree never travels to the GPU box; only the fixtures do.
//
// File format in : int64 N, int64 E, int32 rowptr[N+1], int32 col[E]   (0-based CSR)
// File format out: int64 n_groups, then per group int32 size + int32 ids[size],
//                  then int64 n_order + int32 order[n_order]  (LOI.cpp:873-891 output order)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// Signatures as declared in /root/reference/LOI.cpp:98, :286, :507 and :660.
void reorder_plus(std::vector<int>& row_id, std::vector<int>& col_id, int node_num, std::vector<std::vector<int>>& res,
                  std::vector<bool>& visit);
void reorder_plus_direct(std::vector<int>& row_id, std::vector<int>& col_id, int node_num,
                         std::vector<std::vector<int>>& res, std::vector<bool>& visit, std::vector<int>& row_id_in,
                         std::vector<int>& col_id_in);
void reorder_plus_new(std::vector<int>& row_id, std::vector<int>& col_id, int node_num,
                      std::vector<std::vector<int>>& res, std::vector<bool>& visit);
void reorder_plus_new_direct(std::vector<int>& row_id, std::vector<int>& col_id, int node_num,
                             std::vector<std::vector<int>>& res, std::vector<bool>& visit,
                             std::vector<int>& row_id_in, std::vector<int>& col_id_in);

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <new|new_direct|plus|plus_direct> <in.bin> <out.bin>\n", argv[0]);
    return 2;
  }
  const char* variant = argv[1];
  FILE* fi = std::fopen(argv[2], "rb");
  if (!fi) return 3;
  int64_t N = 0, E = 0;
  if (std::fread(&N, 8, 1, fi) != 1 || std::fread(&E, 8, 1, fi) != 1) return 3;
  std::vector<int> row_id(N + 1), col_id(E);
  if (std::fread(row_id.data(), 4, N + 1, fi) != (size_t)(N + 1)) return 3;
  if (E && std::fread(col_id.data(), 4, E, fi) != (size_t)E) return 3;
  std::fclose(fi);

  // in-CSR by a row-major scan (what LOI.cpp:826-841 builds before the call)
  std::vector<int> row_id_in(N + 1, 0), col_id_in(E);
  for (int64_t e = 0; e < E; ++e) row_id_in[col_id[e] + 1]++;
  for (int64_t i = 0; i < N; ++i) row_id_in[i + 1] += row_id_in[i];
  {
    std::vector<int> fill(row_id_in.begin(), row_id_in.end());
    for (int64_t r = 0; r < N; ++r)
      for (int e = row_id[r]; e < row_id[r + 1]; ++e) col_id_in[fill[col_id[e]]++] = (int)r;
  }

  std::vector<std::vector<int>> res;
  std::vector<bool> visit(N);
  // The reference prints its group counter on every iteration (LOI.cpp:688): keep stdout quiet.
  std::fflush(stdout);
  FILE* quiet = std::freopen("/dev/null", "w", stdout);
  (void)quiet;
  if (!std::strcmp(variant, "new")) reorder_plus_new(row_id, col_id, (int)N, res, visit);
  else if (!std::strcmp(variant, "plus")) reorder_plus(row_id, col_id, (int)N, res, visit);  // windowed (VW = 300)
  else if (!std::strcmp(variant, "plus_direct")) reorder_plus_direct(row_id, col_id, (int)N, res, visit, row_id_in, col_id_in);
  else reorder_plus_new_direct(row_id, col_id, (int)N, res, visit, row_id_in, col_id_in);

  FILE* fo = std::fopen(argv[3], "wb");
  if (!fo) return 4;
  int64_t ng = (int64_t)res.size();
  std::fwrite(&ng, 8, 1, fo);
  for (auto& g : res) {
    int32_t s = (int32_t)g.size();
    std::fwrite(&s, 4, 1, fo);
    std::fwrite(g.data(), 4, g.size(), fo);
  }
  // final order as written by the reference's main (LOI.cpp:873-891)
  std::vector<int> order;
  for (auto& g : res) if (g.size() == 16) order.insert(order.end(), g.begin(), g.end());
  for (auto& g : res) if (g.size() < 16) order.insert(order.end(), g.begin(), g.end());
  for (int64_t i = 0; i < N; ++i) if (!visit[i]) order.push_back((int)i);
  int64_t no = (int64_t)order.size();
  std::fwrite(&no, 8, 1, fo);
  std::fwrite(order.data(), 4, order.size(), fo);
  std::fclose(fo);
  return 0;
}
