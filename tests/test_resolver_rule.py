"""The parallel commit rule of k_list_resolve (openvslam_amd/csrc/match_window.hip) as an executable model, checked against the
sequential loops it replaces (match::projection / area / bow_tree: a query sees the targets earlier queries claimed).

Rule: in a round every pending query evaluates (best, second) against the current claims and stamps every LIVE candidate of its list with
its own index (minimum wins); a query is affected if a lower pending query stamped its best (or its second, where the rule has a ratio
test); every unaffected query commits. The second model below stamps only the candidates a query could still CLAIM (distance within the
rule's threshold) -- tempting, and wrong: a later query then commits a target that an earlier, still pending query has yet to look at
again (as its second-best), so the test also pins that this variant does differ from the sequential result."""
import numpy as np

MAXD = 256


def _accepts(rule, bd, sd, ratio):
    if rule == "best_only":               # match_current_and_last_frames: no ratio test
        return bd <= 100
    if bd > 50:
        return False
    return not (np.float32(ratio) * np.float32(sd) < np.float32(bd))


def _evaluate(cands, thr):
    best = second = None
    bd = sd = MAXD
    for d, t in cands:
        if not d < thr[t]:                # claimed (thr 0) or, area, matched at a distance <= ours
            continue
        if d < bd:
            second, sd, best, bd = best, bd, (d, t), d
        elif d < sd:
            second, sd = (d, t), d
    return best, second, bd, sd


def _commit(rule, q, best, thr, owner, match):
    t = best[1]
    if rule == "area":                    # a closer later query steals the target; the earlier owner loses it
        if owner[t] is not None:
            match.pop(owner[t], None)
        owner[t] = q
        thr[t] = best[0]
    else:
        thr[t] = 0
    match[q] = t


def sequential(rule, lists, n_t, ratio):
    thr, owner, match = [MAXD] * n_t, [None] * n_t, {}
    for q, cands in enumerate(lists):
        best, second, bd, sd = _evaluate(cands, thr)
        if best is not None and _accepts(rule, bd, sd, ratio):
            _commit(rule, q, best, thr, owner, match)
    return match


def parallel(rule, lists, n_t, ratio, batch, max_d, stamp_claimable_only=False):
    thr, owner, match, rounds = [MAXD] * n_t, [None] * n_t, {}, 0
    for q0 in range(0, len(lists), batch):
        pending = {q for q in range(q0, min(q0 + batch, len(lists))) if lists[q]}
        while pending:
            rounds += 1
            mark, ev = {}, {}
            for q in sorted(pending):
                for d, t in lists[q]:
                    if d < thr[t] and (d <= max_d or not stamp_claimable_only):
                        mark[t] = min(mark.get(t, 1 << 30), q)
                ev[q] = _evaluate(lists[q], thr)
            still, commits = set(), []
            for q in sorted(pending):
                best, second, bd, sd = ev[q]
                affected = False
                if best is not None and bd <= max_d:      # a best beyond the threshold is a final reject: it can only grow
                    affected = mark.get(best[1], 1 << 30) < q
                    if rule != "best_only" and second is not None:
                        affected = affected or mark.get(second[1], 1 << 30) < q
                if affected:
                    still.add(q)
                elif best is not None and _accepts(rule, bd, sd, ratio):
                    commits.append((q, best))
            assert len(still) < len(pending)              # the lowest pending query is never affected
            for q, best in commits:
                _commit(rule, q, best, thr, owner, match)
            pending = still
    return match, rounds


def _random_problem(rng):
    n_t, n_q = int(rng.integers(5, 80)), int(rng.integers(5, 120))
    lists = []
    for _ in range(n_q):
        k = int(rng.integers(0, 8))
        ts = rng.choice(n_t, size=min(k, n_t), replace=False)
        lists.append([(int(rng.integers(0, 130)), int(t)) for t in ts])
    return lists, n_t, float(rng.choice([0.6, 0.8, 0.9, 1.0])), int(rng.choice([16, 64]))


def test_parallel_rule_equals_the_sequential_loops():
    rng = np.random.default_rng(1)
    for rule, max_d in (("bow", 50), ("area", 50), ("best_only", 100)):
        for _ in range(700):
            lists, n_t, ratio, batch = _random_problem(rng)
            got, rounds = parallel(rule, lists, n_t, ratio, batch, max_d)
            assert got == sequential(rule, lists, n_t, ratio)
            assert rounds <= len(lists) + len(lists) // batch + 1


def test_stamping_only_claimable_candidates_is_not_exact():
    rng = np.random.default_rng(0)
    differs = 0
    for _ in range(300):
        lists, n_t, ratio, batch = _random_problem(rng)
        got, _ = parallel("bow", lists, n_t, ratio, batch, 50, stamp_claimable_only=True)
        differs += got != sequential("bow", lists, n_t, ratio)
    assert differs > 0



def parallel_prefix(rule, lists, n_t, ratio, batch, max_d):
    """The rule k_bf_resolve (match_hamming.hip) still uses, and k_list_resolve used until round 3: only ACCEPTING queries stamp, and only
    their best; the first affected query cuts the round -- everything below it commits, everything from it on evaluates again."""
    thr, owner, match, rounds = [MAXD] * n_t, [None] * n_t, {}, 0
    for q0 in range(0, len(lists), batch):
        pending = {q for q in range(q0, min(q0 + batch, len(lists))) if lists[q]}
        while pending:
            rounds += 1
            mark, ev = {}, {}
            for q in sorted(pending):
                ev[q] = _evaluate(lists[q], thr)
                best, second, bd, sd = ev[q]
                if best is not None and _accepts(rule, bd, sd, ratio):
                    mark[best[1]] = min(mark.get(best[1], 1 << 30), q)
            first_affected = 1 << 30
            for q in sorted(pending):
                best, second, bd, sd = ev[q]
                if best is not None and bd <= max_d:
                    hit = mark.get(best[1], 1 << 30) < q or (second is not None and mark.get(second[1], 1 << 30) < q)
                    if hit:
                        first_affected = q
                        break
            for q in sorted(pending):
                if q >= first_affected:
                    break
                best, second, bd, sd = ev[q]
                if best is not None and _accepts(rule, bd, sd, ratio):
                    _commit(rule, q, best, thr, owner, match)
            pending = {q for q in pending if q >= first_affected}
    return match, rounds


def _tracked_frame_like_problem(rng):
    """Queries in a random (not spatial) order, each with one to three candidates out of a local neighbourhood of targets: most lists are
    disjoint, a few neighbours share a target -- the shape of the windowed matchers' lists (2.4 candidates per query on the tracked frame)."""
    n_q = int(rng.integers(200, 400))
    n_t = n_q + 50
    home = rng.permutation(n_q)
    lists = []
    for q in range(n_q):
        k = int(rng.integers(1, 4))
        ts = np.unique(np.clip(home[q] + rng.integers(-2, 3, size=k), 0, n_t - 1))
        lists.append([(int(rng.integers(0, 90)), int(t)) for t in ts])
    return lists, n_t, float(rng.choice([0.6, 0.8, 0.9, 1.0])), 64


def test_prefix_rule_is_exact_too_and_which_rule_needs_fewer_rounds():
    """Both rules reproduce the sequential loops. Which one needs fewer rounds depends on how much the lists overlap: with sparse overlap
    (the windowed matchers) committing every unaffected query wins; with dense overlap (random lists over few targets -- and the
    brute-force matcher's near lists, where the device measurement went 0.129 -> 0.213 ms) the wider stamps make more queries wait."""
    rng = np.random.default_rng(7)
    dense = {"prefix": 0, "all": 0}
    sparse = {"prefix": 0, "all": 0}
    for rule, max_d in (("bow", 50), ("area", 50), ("best_only", 100)):
        for k in range(300):
            for gen, acc in ((_random_problem, dense), (_tracked_frame_like_problem, sparse)):
                if gen is _tracked_frame_like_problem and k >= 60:
                    continue
                lists, n_t, ratio, batch = gen(rng)
                want = sequential(rule, lists, n_t, ratio)
                got, rp = parallel_prefix(rule, lists, n_t, ratio, batch, max_d)
                assert got == want
                got2, ra = parallel(rule, lists, n_t, ratio, batch, max_d)
                assert got2 == want
                acc["prefix"] += rp
                acc["all"] += ra
    assert sparse["all"] < sparse["prefix"]
    assert dense["all"] > dense["prefix"]


def parallel_union(rule, lists, n_t, ratio, batch, max_d):
    """A candidate for the next version of both resolvers (not in the kernels yet): commit the UNION of the two safe sets -- every query
    below the first one affected in the prefix rule's sense (a lower accepting query stamped its best / second), and every query no lower
    pending query can interfere with (best not in any lower pending query's live list, second not among their claimable candidates).
    Final rejects do not stamp. Never more rounds than either rule alone."""
    thr, owner, match, rounds = [MAXD] * n_t, [None] * n_t, {}, 0
    for q0 in range(0, len(lists), batch):
        pending = {q for q in range(q0, min(q0 + batch, len(lists))) if lists[q]}
        while pending:
            rounds += 1
            live, claimable, accepted, ev = {}, {}, {}, {}
            for q in sorted(pending):
                ev[q] = _evaluate(lists[q], thr)
                best, second, bd, sd = ev[q]
                if best is None or bd > max_d:
                    continue
                if _accepts(rule, bd, sd, ratio):
                    accepted[best[1]] = min(accepted.get(best[1], 1 << 30), q)
                for d, t in lists[q]:
                    if d < thr[t]:
                        live[t] = min(live.get(t, 1 << 30), q)
                        if d <= max_d:
                            claimable[t] = min(claimable.get(t, 1 << 30), q)
            first = 1 << 30
            for q in sorted(pending):
                best, second, bd, sd = ev[q]
                if best is not None and bd <= max_d and (accepted.get(best[1], 1 << 30) < q or
                                                         (second is not None and accepted.get(second[1], 1 << 30) < q)):
                    first = q
                    break
            still, commits = set(), []
            for q in sorted(pending):
                best, second, bd, sd = ev[q]
                blocked = False
                if best is not None and bd <= max_d:
                    blocked = live.get(best[1], 1 << 30) < q
                    if rule != "best_only" and second is not None:
                        blocked = blocked or claimable.get(second[1], 1 << 30) < q
                if blocked and q >= first:
                    still.add(q)
                elif best is not None and _accepts(rule, bd, sd, ratio):
                    commits.append((q, best))
            assert len(still) < len(pending)
            for q, best in commits:
                _commit(rule, q, best, thr, owner, match)
            pending = still
    return match, rounds


def test_union_of_both_rules_is_exact_and_needs_the_fewest_rounds():
    rng = np.random.default_rng(12)
    for gen, n in ((_tracked_frame_like_problem, 40), (_random_problem, 200)):
        for rule, max_d in (("bow", 50), ("area", 50), ("best_only", 100)):
            r_union = r_prefix = r_all = 0
            for _ in range(n):
                lists, n_t, ratio, batch = gen(rng)
                got, r = parallel_union(rule, lists, n_t, ratio, batch, max_d)
                assert got == sequential(rule, lists, n_t, ratio)
                r_union += r
                r_prefix += parallel_prefix(rule, lists, n_t, ratio, batch, max_d)[1]
                r_all += parallel(rule, lists, n_t, ratio, batch, max_d)[1]
            assert r_union <= min(r_prefix, r_all)


def _bow_like_problem(rng):
    """bow_tree's lists: the queries of a vocabulary node all hold the SAME candidate set (the node's frame keypoints), at different distances,
    and true matches are close -- the extreme of contention (tools/fuzz_parity.py --contention on the device)."""
    lists, n_t = [], 0
    for _ in range(int(rng.integers(2, 6))):
        members = list(range(n_t, n_t + int(rng.integers(3, 40))))
        n_t = members[-1] + 1
        for _ in range(int(rng.integers(3, 40))):
            true = int(rng.choice(members))
            lists.append([(int(rng.integers(5, 45)) if t == true else int(rng.integers(30, 140)), t) for t in members])
    return lists, n_t, float(rng.choice([0.6, 0.75, 0.9, 1.0])), int(rng.choice([16, 64]))


def test_rules_on_lists_shared_by_a_whole_node():
    rng = np.random.default_rng(21)
    for _ in range(150):
        lists, n_t, ratio, batch = _bow_like_problem(rng)
        want = sequential("bow", lists, n_t, ratio)
        for rule_fn in (parallel, parallel_prefix, parallel_union):
            got, rounds = rule_fn("bow", lists, n_t, ratio, batch, 50)
            assert got == want
        assert len(want) > 0
