/*
 * ac_oracle.c -- CPU restatement of RokLenarcic/AhoCorasick's matchers.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in ahocorasick_amd/ (the product) may
 * import, link or execute this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU
 * baseline ("port").
 *
 * Parity pinning: the reference is Java and no JDK exists in the build image,
 * so the reference itself cannot be run here (DESIGN.md "Oracle").  This
 * restatement is pinned against the reference's own deterministic test inputs
 * (T/SetTest.java:61-130, T/MatchQueueTest.java:9-57, README examples) whose
 * expected answers come from the brute-force formulas of the reference tests
 * (T/AhoCorasickTest.java:28-38, T/LongestMatchTest.java:30-42,
 * T/WholeWordMatchTest.java:73-90) -- see tests/golden/ and
 * tests/test_oracle_golden.py.  Case-insensitive mode and non-ASCII word
 * characters are exercised by no reference test: parity there is UNPINNED
 * beyond the code reading below (Character.toLowerCase / isLetterOrDigit are
 * inputs: 65536-entry tables supplied by the caller).
 *
 * It is "reference shaped": pointer-linked nodes, open-addressing HashmapNode
 * (FNV-1a, linear probing) vs dense RangeNode chosen by the RangeNodeThreshold
 * rule, BFS fail links with compressed suffix-match links, the depth-first
 * gap fill, a one-automaton-step-per-UTF-16-unit loop and one indirect call
 * per match -- so that timing it is a fair stand-in for the Java CPU path.
 *
 * S/ = src/main/java/com/roklenarcic/util/strings/ of the reference.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define FAM_AC 0        /* S/AhoCorasickSet.java, S/AhoCorasickMap.java       */
#define FAM_LONGEST 1   /* S/LongestMatchSet.java, S/LongestMatchMap.java     */
#define FAM_WHOLEWORD 2 /* S/WholeWordMatchSet.java, S/WholeWordMatchMap.java */
#define FAM_SHORTEST 3  /* S/ShortestMatchSet.java, S/ShortestMatchMap.java   */
#define FAM_WWLONGEST 4 /* S/WholeWordLongestMatchSet.java, S/WholeWordLongestMatchMap.java */

#define ORACLE_OK 0
#define ORACLE_E_ILLEGAL_ARGUMENT (-2) /* java.lang.IllegalArgumentException */
#define ORACLE_E_NOMEM (-3)

typedef struct Node Node;
struct Node {
    /* TrieNode fields: S/AhoCorasickSet.java:498-503, S/LongestMatchSet.java:507-515 */
    Node *defaultTransition;
    Node *failTransition;
    Node *suffixMatch;
    int32_t matchLength;
    int32_t level;
    int32_t value; /* Map flavour: keyword index standing in for T value */
    /* WholeWordLongest: the last keyword on the path that ended at a word boundary (S/WholeWordLongestMatchMap.java:668-672) */
    int32_t failMatchLength, failMatchOffset, failValue;
    /* representation */
    int32_t isRange;
    Node **children;
    /* HashmapNode: S/AhoCorasickSet.java:260-268 */
    uint16_t *keys;
    int32_t capacity; /* keys.length */
    int32_t modulusMask;
    int32_t numEntries;
    /* RangeNode: S/AhoCorasickSet.java:417-421 */
    uint16_t baseChar;
    int32_t size;
    Node *allNext; /* arena list for freeing */
};

typedef struct oracle {
    int family;
    int caseSensitive;
    int mapFlavour; /* 1: the *Map class of the family is restated where its loops differ from the *Set class's
                       (WholeWordLongest, case-insensitive skip loops: oracle_set_map_flavour) */
    Node *root;
    Node *all;
    uint16_t *lower;    /* Character.toLowerCase(char) table (caller supplied) */
    uint8_t *wordChars; /* WordCharacters flags (caller supplied) */
    int64_t nNodes, nRange, nHash;
    /* Thresholder: S/threshold/RangeNodeThreshold.java:7-21 defaults */
    double exponent, linearFactor, maxValue, constantFactor;
} oracle;

/* ---------------------------------------------------------------- nodes */

static Node *new_hashmap_node(oracle *o, int root, int level) {
    /* S/AhoCorasickSet.java:260-272, :505-507 ; S/LongestMatchSet.java:517-520 */
    Node *n = (Node *)calloc(1, sizeof(Node));
    if (!n) return NULL;
    n->children = (Node **)calloc(1, sizeof(Node *));
    n->keys = (uint16_t *)calloc(1, sizeof(uint16_t));
    n->capacity = 1;
    n->modulusMask = 0;
    n->defaultTransition = root ? n : NULL;
    n->level = level;
    n->value = -1;
    n->allNext = o->all;
    o->all = n;
    o->nNodes++;
    return n;
}

/* FNV-1a over high byte then low byte: S/AhoCorasickSet.java:406-410 */
static inline int32_t hm_hash(uint16_t c) {
    const uint32_t HASH_PRIME = 16777619u;
    return (int32_t)((((0x811c9dc5u ^ (uint32_t)(c >> 8)) * HASH_PRIME) ^ (uint32_t)(c & 0xff)) * HASH_PRIME);
}

/* S/AhoCorasickSet.java:275-289 (hashmap) and :451-460 (range) */
static inline Node *get_transition(const Node *n, uint16_t key) {
    if (n->isRange) {
        int32_t idx = (uint16_t)(key - n->baseChar);
        if (idx < n->size) return n->children[idx];
        return n->defaultTransition;
    } else {
        int32_t defaultSlot = hm_hash(key) & n->modulusMask;
        int32_t currentSlot = defaultSlot;
        do {
            if (n->keys[currentSlot] == key) {
                return n->children[currentSlot];
            } else if (n->children[currentSlot] == NULL) {
                return n->defaultTransition;
            } else {
                currentSlot = (currentSlot + 1) & n->modulusMask;
            }
        } while (currentSlot != defaultSlot);
        return n->defaultTransition;
    }
}

/* S/AhoCorasickSet.java:350-376 */
static int hm_enlarge(Node *n) {
    int32_t newCap = n->capacity * 2;
    uint16_t *biggerKeys = (uint16_t *)calloc((size_t)newCap, sizeof(uint16_t));
    Node **biggerChildren = (Node **)calloc((size_t)newCap, sizeof(Node *));
    if (!biggerKeys || !biggerChildren) return -1;
    int32_t biggerMask = newCap - 1;
    for (int32_t i = 0; i < n->capacity; i++) {
        uint16_t key = n->keys[i];
        Node *node = n->children[i];
        if (node != NULL) {
            int32_t defaultSlot = hm_hash(key) & biggerMask;
            int32_t currentSlot = defaultSlot;
            do {
                if (biggerChildren[currentSlot] == NULL) {
                    biggerKeys[currentSlot] = key;
                    biggerChildren[currentSlot] = node;
                    break;
                } else {
                    currentSlot = (currentSlot + 1) & biggerMask;
                }
            } while (currentSlot != defaultSlot);
        }
    }
    free(n->keys);
    free(n->children);
    n->keys = biggerKeys;
    n->children = biggerChildren;
    n->capacity = newCap;
    n->modulusMask = biggerMask;
    return 0;
}

/* S/AhoCorasickSet.java:380-403 (LongestMatchSet passes level+1: :405 there) */
static Node *hm_get_or_add_child(oracle *o, Node *n, uint16_t key) {
    if (n->capacity < 0x10000 &&
        ((n->numEntries >= n->capacity) || (n->numEntries > 16 && ((float)n->numEntries >= (float)n->capacity * 0.90f)))) {
        if (hm_enlarge(n)) return NULL;
    }
    int32_t defaultSlot = hm_hash(key) & n->modulusMask;
    int32_t currentSlot = defaultSlot;
    do {
        if (n->children[currentSlot] == NULL) {
            n->keys[currentSlot] = key;
            Node *newChild = new_hashmap_node(o, 0, n->level + 1);
            n->children[currentSlot] = newChild;
            ++n->numEntries;
            return newChild;
        } else if (n->keys[currentSlot] == key) {
            return n->children[currentSlot];
        } else {
            currentSlot = (currentSlot + 1) & n->modulusMask;
        }
    } while (currentSlot != defaultSlot);
    return NULL; /* IllegalStateException in the reference; unreachable */
}

/* clear(): S/ShortestMatchSet.java:278-283 (hashmap), :447-450 (range) */
static int node_clear(Node *n) {
    if (n->isRange) {
        free(n->children);
        n->children = NULL;
        n->size = 0;
    } else {
        Node **c = (Node **)calloc(1, sizeof(Node *));
        uint16_t *k = (uint16_t *)calloc(1, sizeof(uint16_t));
        if (!c || !k) { free(c); free(k); return -1; }
        free(n->children);
        free(n->keys);
        n->children = c;
        n->keys = k;
        n->capacity = 1;
        n->modulusMask = 0;
        n->numEntries = 0;
    }
    return 0;
}

static inline int node_is_empty(const Node *n) { /* :292-294, :463-465 */
    return n->isRange ? (n->size == 0) : (n->numEntries == 0);
}

/* S/AhoCorasickSet.java:306-320 (hashmap), :479-493 (range) */
static void update_transition(Node *n, uint16_t c, Node *node) {
    if (n->isRange) {
        int32_t idx = (uint16_t)(c - n->baseChar);
        if (idx < n->size && n->children[idx] != NULL) n->children[idx] = node;
    } else {
        int32_t defaultSlot = hm_hash(c) & n->modulusMask;
        int32_t currentSlot = defaultSlot;
        do {
            if (n->children[currentSlot] == NULL) {
                return;
            } else if (n->keys[currentSlot] == c) {
                n->children[currentSlot] = node;
                return;
            } else {
                currentSlot = (currentSlot + 1) & n->modulusMask;
            }
        } while (currentSlot != defaultSlot);
    }
}

/* S/threshold/RangeNodeThreshold.java:23-29 */
static int is_over_threshold(const oracle *o, int nodeSize, int nodeLevel, int keyIntervalSize) {
    if (keyIntervalSize <= 8) return 1;
    int charArrayCost = (nodeSize / 4) + 3;
    return (double)(nodeSize + charArrayCost) >
           (double)keyIntervalSize * (o->maxValue - o->linearFactor / pow(o->constantFactor + nodeLevel, o->exponent));
}

/* HashmapNode.optimizeNode S/AhoCorasickSet.java:323-346 + RangeNode ctor :423-448.
 * WholeWord flavour (S/WholeWordMatchMap.java:416-436, :521-539) has no root
 * self-transition and no prefill. */
static Node *optimize_node(oracle *o, Node *n, int level) {
    if (n->isRange) return n; /* TrieNode.optimizeNode default: :539-541 */
    uint16_t minKey = 0xffff, maxKey = 0;
    int size = n->numEntries;
    for (int32_t i = 0; i < n->capacity; i++) {
        if (n->children[i] != NULL) {
            if (n->keys[i] > maxKey) maxKey = n->keys[i];
            if (n->keys[i] < minKey) minKey = n->keys[i];
        }
    }
    int keyIntervalSize = (int)maxKey - (int)minKey + 1;
    int isRoot = (n->defaultTransition != NULL);
    if (!(isRoot || is_over_threshold(o, size, level, keyIntervalSize))) return n;

    Node *r = (Node *)calloc(1, sizeof(Node));
    if (!r) return NULL;
    r->isRange = 1;
    r->defaultTransition = isRoot ? r : NULL;
    r->level = n->level;
    r->baseChar = minKey;
    r->size = (int)maxKey - (int)minKey + 1;
    r->matchLength = n->matchLength;
    r->value = n->value;
    if (r->size <= 0) {
        r->size = 0;
    } else {
        r->children = (Node **)calloc((size_t)r->size, sizeof(Node *));
        if (!r->children) { free(r); return NULL; }
        if (isRoot)
            for (int32_t i = 0; i < r->size; i++) r->children[i] = r; /* Arrays.fill(children, this) :438-440 */
        for (int32_t i = 0; i < n->capacity; i++)
            if (n->children[i] != NULL) r->children[n->keys[i] - minKey] = n->children[i];
    }
    r->allNext = o->all;
    o->all = r;
    o->nNodes++;
    return r;
}

/* ------------------------------------------------ S/Queue.java (ring deque) */

typedef struct {
    Node **arr;
    int64_t cap, first, last;
} NQueue;

static int q_init(NQueue *q) {
    q->cap = 50;
    q->arr = (Node **)calloc((size_t)q->cap, sizeof(Node *));
    q->first = q->last = 0;
    return q->arr ? 0 : -1;
}
static int q_empty(const NQueue *q) { return q->first == q->last; }
static int q_push(NQueue *q, Node *n) { /* S/Queue.java:37-56 */
    if (((q->last + 1) % q->cap) == q->first) {
        int64_t newCap = q->cap + (q->cap >> 1);
        Node **na = (Node **)calloc((size_t)newCap, sizeof(Node *));
        if (!na) return -1;
        if (q->first <= q->last) {
            memcpy(na + q->first, q->arr + q->first, (size_t)(q->last - q->first) * sizeof(Node *));
        } else {
            memcpy(na, q->arr + q->first, (size_t)(q->cap - q->first) * sizeof(Node *));
            memcpy(na + (q->cap - q->first), q->arr, (size_t)q->last * sizeof(Node *));
            q->last += q->cap - q->first;
            q->first = 0;
        }
        free(q->arr);
        q->arr = na;
        q->cap = newCap;
    }
    q->arr[q->last] = n;
    q->last = (q->last + 1) % q->cap;
    return 0;
}
static Node *q_take(NQueue *q) { /* FIFO: S/Queue.java:58-67 */
    if (q_empty(q)) return NULL;
    Node *r = q->arr[q->first];
    q->first = (q->first + 1) % q->cap;
    return r;
}
static Node *q_pop(NQueue *q) { /* LIFO: S/Queue.java:26-35 */
    if (q_empty(q)) return NULL;
    if (--q->last < 0) q->last = q->cap - 1;
    return q->arr[q->last];
}

/* ------------------------------------------------------------ construction */

/* The BFS visitor body: S/AhoCorasickSet.java:58-127 (Map: value inheritance
 * S/AhoCorasickMap.java:129-133; Longest identical, S/LongestMatchSet.java:59-127). */
static int visit_fail_and_outputs(oracle *o, NQueue *queue, Node *parent, uint16_t key, Node *value, int level) {
    Node *opt = optimize_node(o, value, level);
    if (!opt) return -1;
    value = opt;
    update_transition(parent, key, value);

    if (o->family == FAM_WWLONGEST) {
        /* S/WholeWordLongestMatchSet.java:226-244 (Map: S/WholeWordLongestMatchMap.java:366-384): carry the last
         * keyword that ended at a word boundary down the path, with its distance from the node. */
        if (parent->matchLength != 0 && !o->wordChars[key]) {
            value->failMatchLength = parent->matchLength;
            value->failMatchOffset = 1;
            value->failValue = parent->value;
        } else {
            value->failMatchLength = parent->failMatchLength;
            value->failMatchOffset = parent->failMatchOffset + 1;
            value->failValue = parent->failValue;
        }
        if (!node_is_empty(value))
            if (q_push(queue, value)) return -1;
        return 0;
    }

    Node *parentFail = parent->failTransition;
    if (parentFail == NULL) {
        value->failTransition = parent; /* depth-1 nodes fail to root :66-70 */
    } else {
        do {
            Node *matchContinuation = get_transition(parentFail, key);
            if (matchContinuation != NULL) {
                value->failTransition = matchContinuation;
            } else {
                parentFail = parentFail->failTransition;
            }
        } while (value->failTransition == NULL);
        if (o->family == FAM_SHORTEST) {
            /* S/ShortestMatchSet.java:104-118 (Map also copies the value, S/ShortestMatchMap.java:112-125): a node
             * without an own match takes the nearest fail ancestor's; a node with any match loses its transitions. */
            if (value->matchLength == 0) {
                Node *fail = value->failTransition;
                while (fail != o->root && fail->matchLength == 0) fail = fail->failTransition;
                value->matchLength = fail->matchLength;
                value->value = fail->value;
            }
            if (value->matchLength != 0) {
                if (node_clear(value)) return -1;
                value->failTransition = o->root;
            }
            if (!node_is_empty(value))
                if (q_push(queue, value)) return -1;
            return 0;
        }
        /* output compression :110-121 */
        Node *fail = value->failTransition;
        while (fail != o->root && fail->matchLength == 0) fail = fail->failTransition;
        if (fail->matchLength > 0) {
            if (value->matchLength == 0) {
                value->matchLength = fail->matchLength;
                value->suffixMatch = fail->suffixMatch;
                value->value = fail->value;
            } else {
                value->suffixMatch = fail;
            }
        }
    }
    if (!node_is_empty(value))
        if (q_push(queue, value)) return -1;
    return 0;
}

/* node.mapEntries(visitor) for the two visitors used by the AC-family ctor.
 * which==0: failTransAndOutputsVisitor; which==1: enqueueNodesVisitor (:147-156).
 * Iteration order follows the reference: slot order for HashmapNode (:297-303),
 * index order for RangeNode skipping self-loops (:468-476). */
static int map_entries(oracle *o, NQueue *queue, Node *n, int which, int level) {
    if (n->isRange) {
        if (n->children != NULL) {
            for (int32_t i = 0; i < n->size; i++) {
                Node *c = n->children[i];
                if (c != NULL && c != n) {
                    if (which == 0) {
                        if (visit_fail_and_outputs(o, queue, n, (uint16_t)(n->baseChar + i), c, level)) return -1;
                    } else if (!node_is_empty(c)) {
                        if (q_push(queue, c)) return -1;
                    }
                }
            }
        }
    } else {
        for (int32_t i = 0; i < n->capacity; i++) {
            Node *c = n->children[i];
            if (c != NULL) {
                if (which == 0) {
                    if (visit_fail_and_outputs(o, queue, n, n->keys[i], c, level)) return -1;
                } else if (!node_is_empty(c)) {
                    if (q_push(queue, c)) return -1;
                }
            }
        }
    }
    return 0;
}

/* WordCharacters.trim: S/WordCharacters.java:41-62. Returns [*ws,*we). */
static void wc_trim(const uint8_t *wordChars, const uint16_t *kw, int64_t len, int64_t *ws, int64_t *we) {
    int64_t wordStart = 0, wordEnd = len;
    for (int64_t i = 0; i < len; i++) {
        if (wordChars[kw[i]]) { wordStart = i; break; }
    }
    for (int64_t i = len - 1; i >= 0; i--) {
        if (wordChars[kw[i]]) { wordEnd = i + 1; break; }
    }
    *ws = wordStart;
    *we = wordEnd;
}

void oracle_free(oracle *o);

/*
 * Build.  kw/off: n_kw keywords as UTF-16 units, keyword i = kw[off[i]..off[i+1]).
 * A Java null keyword and "" are both skipped by the reference
 * (S/AhoCorasickSet.java:27), so both are represented by an empty range.
 * value of keyword i is its index i (Map flavour; last duplicate wins,
 * S/AhoCorasickMap.java:49-50).
 * On ORACLE_E_ILLEGAL_ARGUMENT *err_kw is the index of the offending keyword
 * (S/WholeWordMatchMap.java:263-267).
 */
int oracle_build(int family, const uint16_t *kw, const uint64_t *off, uint32_t n_kw, int caseSensitive,
                 const uint16_t *lower, const uint8_t *wordChars, oracle **out, int64_t *err_kw) {
    oracle *o = (oracle *)calloc(1, sizeof(oracle));
    if (!o) return ORACLE_E_NOMEM;
    o->family = family;
    o->caseSensitive = caseSensitive;
    o->exponent = 1; o->linearFactor = 1; o->maxValue = 0.65; o->constantFactor = 2;
    o->lower = (uint16_t *)malloc(65536 * sizeof(uint16_t));
    if (!o->lower) { oracle_free(o); return ORACLE_E_NOMEM; }
    if (lower) memcpy(o->lower, lower, 65536 * sizeof(uint16_t));
    else for (int i = 0; i < 65536; i++) o->lower[i] = (uint16_t)i;
    if (family == FAM_WHOLEWORD || family == FAM_WWLONGEST) {
        if (!wordChars) { oracle_free(o); return ORACLE_E_ILLEGAL_ARGUMENT; }
        o->wordChars = (uint8_t *)malloc(65536);
        if (!o->wordChars) { oracle_free(o); return ORACLE_E_NOMEM; }
        memcpy(o->wordChars, wordChars, 65536);
    }

    /* root: HashmapNode(true) for AC/Longest (:22), plain HashmapNode for WholeWord (S/WholeWordMatchMap.java:254) */
    o->root = new_hashmap_node(o, family != FAM_WHOLEWORD && family != FAM_WWLONGEST, 0);
    if (!o->root) { oracle_free(o); return ORACLE_E_NOMEM; }

    for (uint32_t k = 0; k < n_kw; k++) {
        const uint16_t *w = kw + off[k];
        int64_t len = (int64_t)(off[k + 1] - off[k]);
        int64_t ws = 0, we = len;
        if (family == FAM_WWLONGEST) {
            /* S/WholeWordLongestMatchSet.java:190-206: trim only -- non-word characters inside a keyword are allowed */
            wc_trim(o->wordChars, w, len, &ws, &we);
        }
        if (family == FAM_WHOLEWORD) {
            /* S/WholeWordMatchMap.java:259-272: trim, validate un-folded chars, skip empty */
            wc_trim(o->wordChars, w, len, &ws, &we);
            for (int64_t i = ws; i < we; i++) {
                if (!o->wordChars[w[i]]) {
                    if (err_kw) *err_kw = (int64_t)k;
                    oracle_free(o);
                    return ORACLE_E_ILLEGAL_ARGUMENT;
                }
            }
        }
        if (we - ws > 0) {
            Node *cur = o->root;
            int shadowed = 0;
            for (int64_t i = ws; i < we; i++) {
                uint16_t c = caseSensitive ? w[i] : o->lower[w[i]];
                cur = hm_get_or_add_child(o, cur, c);
                if (!cur) { oracle_free(o); return ORACLE_E_NOMEM; }
                /* S/ShortestMatchSet.java:33-37: a keyword already on the path (or this same keyword) matches first */
                if (family == FAM_SHORTEST && cur->matchLength != 0) { shadowed = 1; break; }
            }
            if (shadowed) continue;
            cur->matchLength = (int32_t)(we - ws);
            cur->value = (int32_t)k;
        }
    }

    NQueue queue;
    if (q_init(&queue)) { oracle_free(o); return ORACLE_E_NOMEM; }
    int rc = 0;

    if (family == FAM_WHOLEWORD) {
        /* S/WholeWordMatchMap.java:293-321: only root and its children get optimized,
         * the visitor never enqueues. */
        Node *r = optimize_node(o, o->root, 0);
        if (!r) { rc = -1; goto done; }
        o->root = r;
        if (r->isRange) {
            for (int32_t i = 0; i < r->size && !rc; i++) {
                Node *c = r->children ? r->children[i] : NULL;
                if (c != NULL && c != r) {
                    Node *oc = optimize_node(o, c, 1);
                    if (!oc) rc = -1; else update_transition(r, (uint16_t)(r->baseChar + i), oc);
                }
            }
        } else {
            for (int32_t i = 0; i < r->capacity && !rc; i++) {
                Node *c = r->children[i];
                if (c != NULL) {
                    Node *oc = optimize_node(o, c, 1);
                    if (!oc) rc = -1; else update_transition(r, r->keys[i], oc);
                }
            }
        }
        goto done;
    }

    /* BFS: S/AhoCorasickSet.java:49-54, :130-140 */
    {
        Node *r = optimize_node(o, o->root, 0);
        if (!r) { rc = -1; goto done; }
        o->root = r;
        if (q_push(&queue, o->root) || q_push(&queue, NULL)) { rc = -1; goto done; }
        int level = 1;
        while (!q_empty(&queue)) {
            Node *n = q_take(&queue);
            if (n == NULL) {
                if (!q_empty(&queue)) {
                    if (q_push(&queue, NULL)) { rc = -1; goto done; }
                    level++;
                }
            } else {
                if (map_entries(o, &queue, n, 0, level)) { rc = -1; goto done; }
            }
        }
        if (family == FAM_WWLONGEST) goto done; /* a plain trie: no fail transitions, no gap fill */
        /* Depth-first gap fill, restated literally (S/AhoCorasickSet.java:157-190),
         * including the way pop()/push(null) overwrites the popped node's slot;
         * which RangeNodes end up filled is results-neutral (A.2 of SURVEY.md). */
        if (map_entries(o, &queue, o->root, 1, 0)) { rc = -1; goto done; }
        while (!q_empty(&queue)) {
            Node *node = q_pop(&queue);
            if (node == NULL) {
                node = q_pop(&queue);
                if (node != NULL && node->isRange) {
                    for (int32_t i = 0; i < node->size; i++) {
                        if (node->children[i] == NULL) {
                            uint16_t ch = (uint16_t)(node->baseChar + i);
                            Node *n = node->failTransition;
                            while (n != NULL) {
                                Node *nextNode = get_transition(n, ch);
                                if (nextNode == NULL) {
                                    n = n->failTransition;
                                } else {
                                    node->children[i] = nextNode;
                                    break;
                                }
                            }
                        }
                    }
                }
            } else {
                if (q_push(&queue, NULL)) { rc = -1; goto done; }
                if (map_entries(o, &queue, node, 1, 0)) { rc = -1; goto done; }
            }
        }
    }
done:
    free(queue.arr);
    if (rc) { oracle_free(o); return ORACLE_E_NOMEM; }
    for (Node *n = o->all; n; n = n->allNext) { if (n->isRange) o->nRange++; else o->nHash++; }
    *out = o;
    return ORACLE_OK;
}

void oracle_free(oracle *o) {
    if (!o) return;
    Node *n = o->all;
    while (n) {
        Node *nx = n->allNext;
        free(n->children);
        free(n->keys);
        free(n);
        n = nx;
    }
    free(o->lower);
    free(o->wordChars);
    free(o);
}

/* ------------------------------------------------------------- listeners */

/* SetMatchListener/MapMatchListener (S/SetMatchListener.java:6, S/MapMatchListener.java:6):
 * returns nonzero to continue. */
typedef int (*match_listener)(void *ctx, int32_t start, int32_t end, int32_t value);

/* ------------------------------- S/SetMatchQueue.java / S/MapMatchQueue.java */

typedef struct {
    int32_t emptySlotIdx;
    int32_t cap;
    int32_t *endIndexes, *startIndexes, *values;
} MatchQueue;

MatchQueue *oracle_queue_new(void) {
    MatchQueue *q = (MatchQueue *)calloc(1, sizeof(MatchQueue));
    q->cap = 2;
    q->endIndexes = (int32_t *)calloc(2, sizeof(int32_t));
    q->startIndexes = (int32_t *)calloc(2, sizeof(int32_t));
    q->values = (int32_t *)calloc(2, sizeof(int32_t));
    return q;
}
void oracle_queue_free(MatchQueue *q) {
    if (!q) return;
    free(q->endIndexes); free(q->startIndexes); free(q->values); free(q);
}

/* matchAndClear: S/SetMatchQueue.java:19-42 */
static int mq_match_and_clear(MatchQueue *q, match_listener l, void *ctx, int32_t purgeToIndex) {
    if (q->emptySlotIdx != 0) {
        int32_t i = 0;
        while (i < q->emptySlotIdx) {
            if (q->endIndexes[i] <= purgeToIndex) {
                if (!l(ctx, q->startIndexes[i], q->endIndexes[i], q->values[i])) return 0;
            } else {
                break;
            }
            i++;
        }
        if (i > 0) {
            q->emptySlotIdx -= i;
            memmove(q->endIndexes, q->endIndexes + i, (size_t)q->emptySlotIdx * sizeof(int32_t));
            memmove(q->startIndexes, q->startIndexes + i, (size_t)q->emptySlotIdx * sizeof(int32_t));
            memmove(q->values, q->values + i, (size_t)q->emptySlotIdx * sizeof(int32_t));
        }
    }
    return 1;
}

/* push: S/SetMatchQueue.java:45-95 (Map twin carries the value, S/MapMatchQueue.java:75-132) */
static int mq_push(MatchQueue *q, int32_t length, int32_t idx, int32_t value) {
    if (q->emptySlotIdx + 1 == q->cap) {
        int32_t newCap = q->cap * 2;
        q->endIndexes = (int32_t *)realloc(q->endIndexes, (size_t)newCap * sizeof(int32_t));
        q->startIndexes = (int32_t *)realloc(q->startIndexes, (size_t)newCap * sizeof(int32_t));
        q->values = (int32_t *)realloc(q->values, (size_t)newCap * sizeof(int32_t));
        q->cap = newCap;
    }
    if (q->emptySlotIdx != 0) {
        int32_t idxToFind = idx - length;
        for (int32_t currSlot = q->emptySlotIdx - 1; currSlot >= 0; currSlot--) {
            int32_t currStartIdx = q->startIndexes[currSlot];
            if (idxToFind >= currStartIdx) {
                if (idxToFind >= q->endIndexes[currSlot]) {
                    q->startIndexes[currSlot + 1] = idxToFind;
                    q->endIndexes[currSlot + 1] = idx;
                    q->values[currSlot + 1] = value;
                    q->emptySlotIdx = currSlot + 2;
                    return 1;
                } else if (idxToFind == currStartIdx && q->endIndexes[currSlot] < idx) {
                    q->startIndexes[currSlot] = idxToFind;
                    q->endIndexes[currSlot] = idx;
                    q->values[currSlot] = value;
                    q->emptySlotIdx = currSlot + 1;
                    return 1;
                } else {
                    return 0;
                }
            }
        }
        q->startIndexes[0] = idxToFind;
        q->endIndexes[0] = idx;
        q->values[0] = value;
        q->emptySlotIdx = 1;
        return 1;
    } else {
        q->startIndexes[q->emptySlotIdx] = idx - length;
        q->endIndexes[q->emptySlotIdx] = idx;
        q->values[q->emptySlotIdx] = value;
        q->emptySlotIdx++;
        return 1;
    }
}

/* ----------------------------------------------------------------- match */

/* TrieNode.output: S/AhoCorasickSet.java:522-535 / S/AhoCorasickMap.java:627-640 */
static inline int ac_output(const Node *n, match_listener l, void *ctx, int32_t idx) {
    int ret = 1;
    if (n->matchLength > 0) {
        ret = l(ctx, idx - n->matchLength, idx, n->value);
        const Node *sm = n->suffixMatch;
        while (sm != NULL && ret) {
            ret = l(ctx, idx - sm->matchLength, idx, sm->value);
            sm = sm->suffixMatch;
        }
    }
    return ret;
}

/* TrieNode.output(queue, idx): S/LongestMatchSet.java:535-551 */
static inline void longest_output(const Node *n, MatchQueue *q, int32_t idx) {
    int matchAccepted = 0;
    if (n->matchLength != 0) {
        matchAccepted = mq_push(q, n->matchLength, idx, n->value);
        const Node *sm = n->suffixMatch;
        while (sm != NULL && !matchAccepted) {
            matchAccepted = mq_push(q, sm->matchLength, idx, sm->value);
            sm = sm->suffixMatch;
        }
    }
}

/* AhoCorasickSet.match: S/AhoCorasickSet.java:193-252 (Map: S/AhoCorasickMap.java:277-336) */
static void match_ac(const oracle *o, const uint16_t *hay, int32_t len, match_listener l, void *ctx) {
    const Node *currentNode = o->root;
    int32_t idx = 0;
    if (o->caseSensitive) {
        while (idx < len) {
            const uint16_t c = hay[idx];
            const Node *nextNode = get_transition(currentNode, c);
            while (nextNode == NULL) {
                currentNode = currentNode->failTransition;
                nextNode = get_transition(currentNode, c);
            }
            currentNode = nextNode;
            if (!ac_output(currentNode, l, ctx, ++idx)) break;
        }
    } else {
        const uint16_t *lower = o->lower;
        while (idx < len) {
            const uint16_t c = lower[hay[idx]];
            const Node *nextNode = get_transition(currentNode, c);
            while (nextNode == NULL) {
                currentNode = currentNode->failTransition;
                nextNode = get_transition(currentNode, c);
            }
            currentNode = nextNode;
            if (!ac_output(currentNode, l, ctx, ++idx)) break;
        }
    }
}

/* LongestMatchSet.match: S/LongestMatchSet.java:192-265 (Map: S/LongestMatchMap.java:288-360) */
static void match_longest(const oracle *o, const uint16_t *hay, int32_t len, match_listener l, void *ctx) {
    const Node *currentNode = o->root;
    MatchQueue *queue = oracle_queue_new(); /* per call: :196 */
    int32_t idx = 0;
    const uint16_t *lower = o->lower;
    const int cs = o->caseSensitive;
    while (idx < len) {
        const uint16_t c = cs ? hay[idx] : lower[hay[idx]];
        const Node *nextNode = get_transition(currentNode, c);
        int failTransition = 0;
        while (nextNode == NULL) {
            failTransition = 1;
            currentNode = currentNode->failTransition;
            nextNode = get_transition(currentNode, c);
        }
        currentNode = nextNode;
        longest_output(currentNode, queue, ++idx);
        if (failTransition && !mq_match_and_clear(queue, l, ctx, idx - currentNode->level)) {
            oracle_queue_free(queue);
            return;
        }
    }
    mq_match_and_clear(queue, l, ctx, INT32_MAX);
    oracle_queue_free(queue);
}

/* WholeWordMatchMap.match(String): S/WholeWordMatchMap.java:155-240
 * (Set: S/WholeWordMatchSet.java:47-132).  CI: the transition and the
 * !wordChars[c] test see the folded unit (:204,:209), the two skip loops the
 * raw unit (:221,:226). */
static void match_wholeword(const oracle *o, const uint16_t *hay, int32_t len, match_listener l, void *ctx) {
    const Node *root = o->root;
    const Node *currentNode = root;
    const uint8_t *wordChars = o->wordChars;
    const uint16_t *lower = o->lower;
    const int cs = o->caseSensitive;
    int32_t idx = 0;
    while (idx < len) {
        uint16_t c = cs ? hay[idx] : lower[hay[idx]];
        const Node *nextNode = get_transition(currentNode, c);
        if (nextNode == NULL) {
            if (!wordChars[c]) {
                if (currentNode->matchLength != 0) {
                    if (!l(ctx, idx - currentNode->matchLength, idx, currentNode->value)) return;
                }
            } else {
                while (++idx < len && wordChars[hay[idx]]) {
                }
            }
            while (++idx < len && !wordChars[hay[idx]]) {
            }
            currentNode = root;
        } else {
            ++idx;
            currentNode = nextNode;
        }
    }
    if (currentNode->matchLength != 0) {
        l(ctx, idx - currentNode->matchLength, idx, currentNode->value);
    }
}

/* ---- match(Readable, ReadableMatchListener<T>): the haystack arrives through a CharBuffer of charBufferSize units.
 * The listener only receives the value (S/ReadableMatchListener.java:7).  TEST ANNOTATION: the two word-matcher loops
 * restated below also hand the collecting listener the positions the String loops' arithmetic gives at the same place
 * (idx = global position of the unit just read), so that the positions acgpu_stream_feed returns can be checked too.
 * AhoCorasickMap.match(Readable) (S/AhoCorasickMap.java:208-275) and LongestMatchMap.match(Readable)
 * (S/LongestMatchMap.java:203-286) are the String loops with buf.get() in place of charAt(idx), so they are
 * restated by calling those loops; WholeWordMatchMap.match(Readable) (S/WholeWordMatchMap.java:55-153) has its own
 * shape -- scroll() (:325-339) with FOLDED word-character lookups in case-insensitive mode -- restated literally
 * below, including the buffer refills. */
typedef struct {
    const uint16_t *hay;
    int32_t len, next;      /* the Readable: next unit to hand out */
    int32_t cap;            /* CharBuffer capacity */
    int32_t base, pos, lim; /* buffer window [base, base+lim) of hay, position pos (relative) */
} CharBuf;

static int cb_read(CharBuf *b) { /* haystack.read(buf) after buf.clear(), then buf.flip(); -1 at the end */
    if (b->next >= b->len) return -1;
    int32_t n = b->len - b->next < b->cap ? b->len - b->next : b->cap;
    b->base = b->next;
    b->next += n;
    b->pos = 0;
    b->lim = n;
    return n;
}

/* scroll: S/WholeWordMatchMap.java:325-339.  Returns 1 at the end of the haystack. */
static int ww_scroll(const oracle *o, CharBuf *b, int wordChars) {
    for (;;) {
        while (b->pos < b->lim) {
            uint16_t c = b->hay[b->base + b->pos++];
            if (!o->caseSensitive) c = o->lower[c];
            if ((o->wordChars[c] != 0) != (wordChars != 0)) {
                b->pos--;
                return 0;
            }
        }
        if (cb_read(b) == -1) return 1;
    }
}

static void match_wholeword_readable(const oracle *o, const uint16_t *hay, int32_t len, int32_t bufsize, match_listener l,
                                     void *ctx) {
    const Node *root = o->root;
    const Node *currentNode = root;
    CharBuf b = {hay, len, 0, bufsize, 0, 0, 0};
    int done = 0;
    while (!done && cb_read(&b) != -1) {
        while (b.pos < b.lim) {
            uint16_t c = b.hay[b.base + b.pos++];
            if (!o->caseSensitive) c = o->lower[c];
            const Node *nextNode = get_transition(currentNode, c);
            if (nextNode == NULL) {
                if (!o->wordChars[c]) {
                    if (currentNode->matchLength != 0) {
                        const int32_t idx = b.base + b.pos - 1; /* position of c (test annotation, see above) */
                        if (!l(ctx, idx - currentNode->matchLength, idx, currentNode->value)) return;
                    }
                } else {
                    if (ww_scroll(o, &b, 1)) {
                        currentNode = root;
                        done = 1;
                        break;
                    }
                }
                currentNode = root;
                if (ww_scroll(o, &b, 0)) {
                    done = 1;
                    break;
                }
            } else {
                currentNode = nextNode;
            }
        }
    }
    if (currentNode->matchLength != 0) l(ctx, len - currentNode->matchLength, len, currentNode->value);
}

/* WholeWordLongestMatchMap.match(Readable, ReadableMatchListener<T>): S/WholeWordLongestMatchMap.java:54-181, with scroll()
 * :401-415 (the same shape as WholeWordMatchMap's: folded word-character lookups in case-insensitive mode). */
static void match_wwlongest_readable(const oracle *o, const uint16_t *hay, int32_t len, int32_t bufsize, match_listener l,
                                     void *ctx) {
    const Node *root = o->root;
    const Node *currentNode = root;
    CharBuf b = {hay, len, 0, bufsize, 0, 0, 0};
    int done = 0;
    while (!done && cb_read(&b) != -1) {
        while (b.pos < b.lim) {
            uint16_t c = b.hay[b.base + b.pos++];
            if (!o->caseSensitive) c = o->lower[c];
            const Node *nextNode = get_transition(currentNode, c);
            if (nextNode == NULL) {
                if (!o->wordChars[c]) {
                    const int32_t idx = b.base + b.pos - 1; /* position of c (test annotation) */
                    if (currentNode->matchLength != 0) {
                        if (!l(ctx, idx - currentNode->matchLength, idx, currentNode->value)) return;
                    } else if (currentNode->failMatchLength != 0) {
                        const int32_t fe = idx - currentNode->failMatchOffset;
                        if (!l(ctx, fe - currentNode->failMatchLength, fe, currentNode->failValue)) return;
                    }
                } else {
                    if (currentNode->failMatchLength != 0) {
                        const int32_t fe = b.base + b.pos - 1 - currentNode->failMatchOffset;
                        if (!l(ctx, fe - currentNode->failMatchLength, fe, currentNode->failValue)) return;
                    }
                    if (ww_scroll(o, &b, 1)) {
                        currentNode = root;
                        done = 1;
                        break;
                    }
                }
                currentNode = root;
                if (ww_scroll(o, &b, 0)) {
                    done = 1;
                    break;
                }
            } else {
                currentNode = nextNode;
            }
        }
    }
    if (currentNode->matchLength != 0) l(ctx, len - currentNode->matchLength, len, currentNode->value);
    else if (currentNode->failMatchLength != 0) {
        const int32_t fe = len - currentNode->failMatchOffset;
        l(ctx, fe - currentNode->failMatchLength, fe, currentNode->failValue);
    }
}

/* values only, in listener-call order; bufsize = charBufferSize */
int64_t oracle_match_readable(const oracle *o, const uint16_t *hay, int32_t len, int32_t bufsize, int32_t *out, int64_t cap,
                              int64_t stop_after);

/* ShortestMatchSet.match: S/ShortestMatchSet.java:193-262 (Map: S/ShortestMatchMap.java:294-372).  The current node
 * lags one unit behind: a match found on it is reported when the next unit is looked at, then matching restarts at
 * the root with that unit. */
static void match_shortest(const oracle *o, const uint16_t *hay, int32_t len, match_listener l, void *ctx) {
    const Node *root = o->root;
    const Node *currentNode = root;
    int32_t currentNodeMatchLength = currentNode->matchLength;
    int32_t currentNodeMatchValue = currentNode->value;
    int32_t idx = 0;
    while (idx < len) {
        const uint16_t c = o->caseSensitive ? hay[idx] : o->lower[hay[idx]];
        if (currentNodeMatchLength != 0) {
            if (!l(ctx, idx - currentNodeMatchLength, idx, currentNodeMatchValue)) return;
            currentNode = get_transition(root, c);
        } else {
            const Node *nextNode = get_transition(currentNode, c);
            while (nextNode == NULL) {
                currentNode = currentNode->failTransition;
                nextNode = get_transition(currentNode, c);
            }
            currentNode = nextNode;
        }
        currentNodeMatchLength = currentNode->matchLength;
        currentNodeMatchValue = currentNode->value;
        ++idx;
    }
    if (currentNodeMatchLength != 0) l(ctx, idx - currentNodeMatchLength, idx, currentNodeMatchValue);
}

/* WholeWordLongestMatchSet.match: S/WholeWordLongestMatchSet.java:47-178, and the case-sensitive loop of
 * WholeWordLongestMatchMap.match(String) S/WholeWordLongestMatchMap.java:194-245 (the same statements).
 * CI (Set, :120-177): the transition and the !wordChars[c] test see the folded unit (:122,:126), the two skip loops
 * the RAW unit (:151,:156). */
static void match_wwlongest(const oracle *o, const uint16_t *hay, int32_t len, match_listener l, void *ctx) {
    const Node *root = o->root;
    const Node *currentNode = root;
    const uint8_t *wordChars = o->wordChars;
    int32_t idx = 0;
    while (idx < len) {
        const uint16_t c = o->caseSensitive ? hay[idx] : o->lower[hay[idx]];
        const Node *nextNode = get_transition(currentNode, c);
        if (nextNode == NULL) {
            if (!wordChars[c]) {
                if (currentNode->matchLength != 0) {
                    if (!l(ctx, idx - currentNode->matchLength, idx, currentNode->value)) return;
                } else if (currentNode->failMatchLength != 0) {
                    const int32_t failMatchEnd = idx - currentNode->failMatchOffset;
                    if (!l(ctx, failMatchEnd - currentNode->failMatchLength, failMatchEnd, currentNode->failValue)) return;
                }
            } else {
                if (currentNode->failMatchLength != 0) {
                    const int32_t failMatchEnd = idx - currentNode->failMatchOffset;
                    if (!l(ctx, failMatchEnd - currentNode->failMatchLength, failMatchEnd, currentNode->failValue)) return;
                }
                while (++idx < len && wordChars[hay[idx]]) {
                }
            }
            while (++idx < len && !wordChars[hay[idx]]) {
            }
            currentNode = root;
        } else {
            ++idx;
            currentNode = nextNode;
        }
    }
    if (currentNode->matchLength != 0) {
        l(ctx, idx - currentNode->matchLength, idx, currentNode->value);
    } else if (currentNode->failMatchLength != 0) {
        const int32_t failMatchEnd = idx - currentNode->failMatchOffset;
        l(ctx, failMatchEnd - currentNode->failMatchLength, failMatchEnd, currentNode->failValue);
    }
}

/* WholeWordLongestMatchMap.match(String), case-insensitive loop: S/WholeWordLongestMatchMap.java:246-304.  Unlike the
 * Set class it folds in the two skip loops as well (:283, :288: wordChars[Character.toLowerCase(haystack.charAt(idx))]),
 * so with a word-character table that is not fold-consistent the two classes report different matches. */
static void match_wwlongest_map_ci(const oracle *o, const uint16_t *hay, int32_t len, match_listener l, void *ctx) {
    const Node *root = o->root;
    const Node *currentNode = root;
    const uint8_t *wordChars = o->wordChars;
    const uint16_t *lower = o->lower;
    int32_t idx = 0;
    while (idx < len) {
        const uint16_t c = lower[hay[idx]];
        const Node *nextNode = get_transition(currentNode, c);
        if (nextNode == NULL) {
            if (!wordChars[c]) {
                if (currentNode->matchLength != 0) {
                    if (!l(ctx, idx - currentNode->matchLength, idx, currentNode->value)) return;
                } else if (currentNode->failMatchLength != 0) {
                    const int32_t failMatchEnd = idx - currentNode->failMatchOffset;
                    if (!l(ctx, failMatchEnd - currentNode->failMatchLength, failMatchEnd, currentNode->failValue)) return;
                }
            } else {
                if (currentNode->failMatchLength != 0) {
                    const int32_t failMatchEnd = idx - currentNode->failMatchOffset;
                    if (!l(ctx, failMatchEnd - currentNode->failMatchLength, failMatchEnd, currentNode->failValue)) return;
                }
                while (++idx < len && wordChars[lower[hay[idx]]]) {
                }
            }
            while (++idx < len && !wordChars[lower[hay[idx]]]) {
            }
            currentNode = root;
        } else {
            ++idx;
            currentNode = nextNode;
        }
    }
    if (currentNode->matchLength != 0) {
        l(ctx, idx - currentNode->matchLength, idx, currentNode->value);
    } else if (currentNode->failMatchLength != 0) {
        const int32_t failMatchEnd = idx - currentNode->failMatchOffset;
        l(ctx, failMatchEnd - currentNode->failMatchLength, failMatchEnd, currentNode->failValue);
    }
}

static void match_dispatch(const oracle *o, const uint16_t *hay, int32_t len, match_listener l, void *ctx) {
    switch (o->family) {
    case FAM_WWLONGEST:
        if (o->mapFlavour && !o->caseSensitive) match_wwlongest_map_ci(o, hay, len, l, ctx);
        else match_wwlongest(o, hay, len, l, ctx);
        break;
    case FAM_SHORTEST: match_shortest(o, hay, len, l, ctx); break;
    case FAM_AC: match_ac(o, hay, len, l, ctx); break;
    case FAM_LONGEST: match_longest(o, hay, len, l, ctx); break;
    default: match_wholeword(o, hay, len, l, ctx); break;
    }
}

/* A collecting listener that returns false on its stop_after-th call (to pin
 * the early-stop contract, R/README.md:70). */
typedef struct {
    int32_t *out;
    int64_t cap, n, stop_after;
} Collect;

static int collect_listener(void *ctx, int32_t start, int32_t end, int32_t value) {
    Collect *c = (Collect *)ctx;
    if (c->n < c->cap) {
        c->out[3 * c->n + 0] = start;
        c->out[3 * c->n + 1] = end;
        c->out[3 * c->n + 2] = value;
    }
    c->n++;
    return !(c->stop_after >= 0 && c->n >= c->stop_after);
}

/* Runs match(); writes up to cap (start,end,value) triples; returns the number
 * of listener calls made.  stop_after<0: listener always returns true. */
int64_t oracle_match(const oracle *o, const uint16_t *hay, int32_t len, int32_t *out, int64_t cap, int64_t stop_after) {
    Collect c = {out, cap, 0, stop_after};
    match_dispatch(o, hay, len, collect_listener, &c);
    return c.n;
}

int64_t oracle_match_readable(const oracle *o, const uint16_t *hay, int32_t len, int32_t bufsize, int32_t *out, int64_t cap,
                              int64_t stop_after) {
    Collect c = {out, cap, 0, stop_after};
    if (o->family == FAM_WHOLEWORD) match_wholeword_readable(o, hay, len, bufsize > 0 ? bufsize : 1024, collect_listener, &c);
    else if (o->family == FAM_WWLONGEST) match_wwlongest_readable(o, hay, len, bufsize > 0 ? bufsize : 1024, collect_listener, &c);
    else match_dispatch(o, hay, len, collect_listener, &c);
    return c.n;
}

/* The README's "empty listener" (R/README.md:144): one indirect call per match,
 * nothing stored.  Used for the timed CPU baseline. */
static int counting_listener(void *ctx, int32_t start, int32_t end, int32_t value) {
    (void)start; (void)end; (void)value;
    ++*(int64_t *)ctx;
    return 1;
}
static match_listener volatile g_noop = counting_listener;

int64_t oracle_match_count(const oracle *o, const uint16_t *hay, int32_t len) {
    int64_t n = 0;
    match_dispatch(o, hay, len, g_noop, &n);
    return n;
}

/* Which class of a family the String loop restates where Set and Map differ (WholeWordLongest CI): 0 = *Set, 1 = *Map. */
void oracle_set_map_flavour(oracle *o, int map_flavour) { o->mapFlavour = map_flavour != 0; }

int64_t oracle_num_nodes(const oracle *o, int which) {
    return which == 0 ? o->nNodes : which == 1 ? o->nRange : o->nHash;
}

/* ---------- queue hooks for T/MatchQueueTest.java:9-57 (exact sequences) */
int oracle_queue_push(MatchQueue *q, int32_t length, int32_t idx) { return mq_push(q, length, idx, -1); }
int64_t oracle_queue_match_and_clear(MatchQueue *q, int32_t purgeTo, int32_t *out, int64_t cap) {
    Collect c = {out, cap, 0, -1};
    mq_match_and_clear(q, collect_listener, &c, purgeTo);
    return c.n;
}

/* WordCharacters.trim exposed for tests (S/WordCharacters.java:41-62) */
void oracle_trim(const uint8_t *wordChars, const uint16_t *kw, int64_t len, int64_t *ws, int64_t *we) {
    wc_trim(wordChars, kw, len, ws, we);
}
