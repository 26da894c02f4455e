// Mutates a verification key / proof pair byte by byte and runs the compiled verifier (zokrates_amd/csrc/host/verify.cpp) on the
// result: every outcome must be a verdict or a zokrates_hip::Error — never a crash (tests/test_verify.py builds this with the
// address and undefined-behaviour sanitizers) — and a PASSED verdict must come from values equal to the unmutated ones.
// usage: verify_fuzz <verification.key> <proof.json> <iterations> <seed>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <random>
#include "../../include/zkhip_backend.hpp"
using namespace zokrates_hip;
static std::string slurp(const char* p){ std::ifstream f(p); return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
int main(int argc, char** argv){
  std::string vks = slurp(argv[1]), prs = slurp(argv[2]);
  auto vk0 = VerificationKey::from_json(vks); auto pr0 = Proof::from_json(prs);
  std::mt19937_64 g(atoi(argv[4]));
  int ok=0, err=0, pass=0, fail=0;
  for (int it=0; it<atoi(argv[3]); ++it){
    std::string a = vks, b = prs;
    std::string& t = (it&1) ? a : b;
    int nm = 1 + g()%3;
    for (int k=0;k<nm;++k){
      size_t pos = g()%t.size();
      switch (g()%5){
        case 0: t[pos] = (char)(g()%256); break;
        case 1: t.erase(pos, 1 + g()%8); break;
        case 2: t.insert(pos, 1 + g()%4, "[{\"0xa,:]}"[g()%10]); break;
        case 3: t[pos] = "0123456789abcdef"[g()%16]; break;
        case 4: t.resize(pos); break;
      }
      if (t.empty()) t = "x";
    }
    try { auto vk = VerificationKey::from_json(a); auto pr = Proof::from_json(b); bool r = verify(vk, pr); ++ok; (r?pass:fail)++;
      if (r) { auto lower=[](std::string x){ for(auto&c:x) c=tolower(c); return x; };
        auto canon_in=[&](const std::string& x){ std::string y=lower(x); size_t i=0; while(y.compare(i,2,"0x")==0) i+=2; y=y.substr(i); size_t j=y.find_first_not_of('0'); return j==std::string::npos?std::string("0"):y.substr(j); };
        bool same = lower(pr.proof.a.x)==lower(pr0.proof.a.x) && lower(pr.proof.a.y)==lower(pr0.proof.a.y) && lower(pr.proof.c.x)==lower(pr0.proof.c.x) && lower(pr.proof.c.y)==lower(pr0.proof.c.y)
          && lower(pr.proof.b.x[0])==lower(pr0.proof.b.x[0]) && lower(pr.proof.b.x[1])==lower(pr0.proof.b.x[1]) && lower(pr.proof.b.y[0])==lower(pr0.proof.b.y[0]) && lower(pr.proof.b.y[1])==lower(pr0.proof.b.y[1]) && pr.inputs.size()==pr0.inputs.size();
        for (size_t i=0;same && i<pr.inputs.size();++i) same = canon_in(pr.inputs[i])==canon_in(pr0.inputs[i]);
        same = same && vk.query.size()==vk0.query.size();
        for (size_t i=0;same && i<vk.query.size();++i) same = lower(vk.query[i].x)==lower(vk0.query[i].x) && lower(vk.query[i].y)==lower(vk0.query[i].y);
        if (!same) { printf("PASSED WITH DIFFERENT VALUES\n%s\n%s\n", a.c_str(), b.c_str()); return 1; } } }
    catch (const Error& e) { ++err; }
  }
  printf("verified %d (passed %d failed %d), errors %d\n", ok, pass, fail, err);
}
