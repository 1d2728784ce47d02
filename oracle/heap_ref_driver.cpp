// oracle/heap_ref_driver.cpp -- TEST INFRASTRUCTURE.  A driver (this file is
// ours) around the REFERENCE's own header-only, std-only intrusive heap,
// compiled from where it lies: -I/root/reference/smpl/include
// (smpl/include/smpl/intrusive_heap.h + detail/intrusive_heap.hpp).
// Output only goes to oracle/_ref/.  It generates tests/golden/heap_ref_*.json
// through tests/golden/make_heap_golden.py; it never travels as source.
//
// stdin: nops, then nops pairs (code key) -- same op language as
// orc_heap_run in oracle_capi.cpp.  stdout: top element index after each op.
#include <smpl/intrusive_heap.h>

#include <cstdio>
#include <memory>
#include <vector>

struct Elem : public sbpl::heap_element { int prio = 0; int idx = 0; };
struct Less { bool operator()(const Elem& a, const Elem& b) const { return a.prio < b.prio; } };

int main()
{
    int nops = 0;
    if (scanf("%d", &nops) != 1) return 1;
    std::vector<std::unique_ptr<Elem>> elems;
    sbpl::intrusive_heap<Elem, Less> heap;
    for (int i = 0; i < nops; ++i) {
        int code, key;
        if (scanf("%d %d", &code, &key) != 2) return 1;
        if (code == 0) {
            elems.emplace_back(new Elem);
            elems.back()->prio = key;
            elems.back()->idx = (int)elems.size() - 1;
            heap.push(elems.back().get());
        } else if (code == 1) {
            if (!heap.empty()) heap.pop();
        } else if (code == 2 || code == 5) {
            const int e = key >> 20, p = key & 0xFFFFF;
            if (e < (int)elems.size() && heap.contains(elems[e].get())) {
                elems[e]->prio = p;
                if (code == 2) heap.decrease(elems[e].get()); else heap.increase(elems[e].get());
            }
        } else if (code == 3) {
            if (key < (int)elems.size() && heap.contains(elems[key].get())) heap.erase(elems[key].get());
        } else if (code == 4) {
            for (auto it = heap.begin(); it != heap.end(); ++it) (*it)->prio = ((*it)->prio * 7919 + 13) % 1000;
            heap.make();
        }
        printf("%d\n", heap.empty() ? -1 : heap.min()->idx);
    }
    return 0;
}
