/* buildHdpUtil -- drop-in for the reference's executable of the same name (impl/buildHdpUtil.c): builds NanoporeHDP models
 * (.nhdp) from a table of k-mer assignments.  Option table, start-up messages and the order of the work follow the reference
 * (trainModels.py:871-904 composes the command line); the work itself is the library's: sa_hdp_state_new (the model layouts of
 * impl/nanopore_hdp.c:1146-1420 loadNanoporeHdpFromScratch), sa_hdp_nig_params_from_table, sa_hdp_state_pass_assignment_file,
 * sa_hdp_state_gibbs (sweeps on the host, every kept sample's grid evaluation and mixing on the GPU), sa_hdp_state_finalize,
 * sa_hdp_state_write.
 *
 * Beside the reference's options: --seed <n> (the sweeps draw from one seeded generator; default 1), --device <n>, and
 * --updateFrom <in.nhdp> --expectations <file> -v <out.nhdp>: updateHdpFromAssignments (impl/buildHdpUtil.c:66-81,
 * impl/signalMachine.c:384-397), a function the reference carries and never reaches from its main().
 */
#include <getopt.h>
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "signalalign_hip.h"

#define NULL_HYPERPARAMETER -1.0

static void die(const char *msg) {   /* st_errAbort */
    fprintf(stderr, "%s\n", msg);
    exit(1);
}

static void usage(void) {   /* the reference's option letters; the wording is this tool's */
    static const char *const text =
        "\nbuildHdpUtil: build a NanoporeHDP (.nhdp) from a table of k-mer assignments\n\n"
        "  --help            this text\n"
        "  --verbose         one line per Gibbs sweep on stderr\n"
        "  --oneD            1-D reads: template model only\n"
        "  -p <type>         model layout, alphabet and concentration parameters:\n"
        "                      0/1  flat            fixed / Gamma prior   ACEGOT\n"
        "                      2/3  multiset        fixed / Gamma prior   ACEGOT\n"
        "                      4/5  composition     fixed / Gamma prior   ACEGOT\n"
        "                      6/7  middle nts      fixed / Gamma prior   ACEGOT\n"
        "                      8/9  group multiset  fixed / Gamma prior   ACEGOT\n"
        "                      10   flat, prior, ACEGT      11  multiset, prior, ACEGT\n"
        "                      12   multiset, prior, ACEGIT 13  flat, prior, ACEGIT\n"
        "                      14   flat, fixed, ACGT\n"
        "                      15-20 flat, fixed: ACFGT, ACGTbp, ACEGTbdehip, ACGTabcdefghijklm, ACGTabcdefghijklmnopq, ACGTabc\n"
        "                      other flat, fixed, alphabet from -b\n"
        "  -a <k>            k-mer length\n"
        "  -b <letters>      alphabet of a type above 20\n"
        "  -T, -C <file>     template / complement lookup table (signalAlign model format)\n"
        "  -l <file>         assignments (kmer, strand, mean, probability) or a full alignment table\n"
        "  -v, -w <file>     template / complement output\n"
        "  -n, -I, -t <n>    distribution samples to keep, burn-in iterations, thinning\n"
        "  -B, -M, -L <x>    fixed concentration parameter of the base / middle / leaf level\n"
        "  -g, -r <x>        Gamma prior (shape, rate) on the base level's\n"
        "  -j, -y <x>        ... on the middle level's\n"
        "  -i, -u <x>        ... on the leaf level's\n"
        "  -s, -e <x>, -k <n>  sampling grid: start, end, points\n"
        "  --seed <n>        seed of the sweeps' generator (default 1)\n"
        "  --device <n>      GPU that evaluates the distribution samples (default 0)\n"
        "  --updateFrom <in.nhdp> --expectations <file> -v <out.nhdp>\n"
        "                    add an HDP expectations file's assignments to an existing model and sample again\n";
    fputs(text, stderr);
    exit(1);
}

typedef struct {
    int64_t type, kmer_length, n_samples, burn_in, thinning, grid_length;
    double base_gamma, middle_gamma, leaf_gamma, bga, bgb, mga, mgb, lga, lgb, grid_start, grid_end;
    const char *alphabet;
    int verbose, device;
    uint64_t seed;
} opts_t;

/* loadNanoporeHdpFromScratch (impl/nanopore_hdp.c:1146-1420): HDP type -> layout, alphabet, fixed gammas or Gamma priors */
static sa_hdp_state_t *from_scratch(const opts_t *o, const char *model_file) {
    sa_model_t *m = NULL;
    if (sa_model_load(&m, model_file, NULL) != SA_OK) {
        fprintf(stderr, "normal_inverse_gamma_params_from_minION: cannot read the lookup table %s\n", model_file);
        exit(1);
    }
    char m_alpha[64];
    int na = 0, k = 0;
    sa_model_alphabet(m, m_alpha, &na, &k);
    int64_t nk = 1;
    for (int i = 0; i < k; i++) nk *= na;
    double mu, nu, alpha, beta;
    if (sa_hdp_nig_params_from_table(sa_model_table5(m), nk, &mu, &nu, &alpha, &beta) != SA_OK)
        die("MLE estimation of alpha numerically unstable at designated starting value.");
    sa_model_destroy(m);

    int layout = SA_HDP_LAYOUT_FLAT, prior = 0;
    const char *alphabet = "ACEGOT";
    int64_t groups[64];
    const int64_t *gp = NULL;
    switch (o->type) {
        case 0: break;                                                      /* singleLevelFixed            */
        case 1: prior = 1; break;                                           /* singleLevelPrior            */
        case 2: layout = SA_HDP_LAYOUT_MULTISET; break;                     /* multisetFixed               */
        case 3: layout = SA_HDP_LAYOUT_MULTISET; prior = 1; break;          /* multisetPrior               */
        case 4: case 5:                                                     /* compFixed / compPrior: PURINES "AG", PYRIMIDINES "CEOT" */
            layout = SA_HDP_LAYOUT_COMPOSITION; prior = o->type == 5;
            alphabet = "AGCEOT";
            for (int i = 0; i < 6; i++) groups[i] = i < 2;
            gp = groups;
            break;
        case 6: layout = SA_HDP_LAYOUT_MIDDLE_NTS; break;                   /* middleNtsFixed              */
        case 7: layout = SA_HDP_LAYOUT_MIDDLE_NTS; prior = 1; break;        /* middleNtsPrior              */
        case 8: case 9: {                                                   /* groupMultiset*: ACEGOT -> {0, 1, 1, 2, 1, 3} */
            static const int64_t g6[6] = {0, 1, 1, 2, 1, 3};
            layout = SA_HDP_LAYOUT_GROUP_MULTISET; prior = o->type == 9;
            memcpy(groups, g6, sizeof(g6));
            gp = groups;
            break;
        }
        case 10: prior = 1; alphabet = "ACEGT"; break;                      /* singleLevelPrior2           */
        case 11: layout = SA_HDP_LAYOUT_MULTISET; prior = 1; alphabet = "ACEGT"; break;    /* multisetPrior2     */
        case 12: layout = SA_HDP_LAYOUT_MULTISET; prior = 1; alphabet = "ACEGIT"; break;   /* multisetPriorEcoli */
        case 13: prior = 1; alphabet = "ACEGIT"; break;                     /* singleLevelPriorEcoli       */
        case 14: alphabet = "ACGT"; break;                                  /* singleLevelFixedCanonical   */
        /* the flat, fixed-gamma models whose alphabets the reference hard-codes (inc/stateMachine.h:25-30, impl/nanopore_hdp.c:1160-1240) */
        case 15: alphabet = "ACFGT"; break;                                 /* singleLevelFixedM6A        METHYL_ADENOSINE_RNA */
        case 16: alphabet = "ACGTbp"; break;                                /* singleLevelFixedrRNA       M7G_PSI_RRNA         */
        case 17: alphabet = "ACEGTbdehip"; break;                           /* singleLevelAll16SrRNA      ALL_16SRRNA          */
        case 18: alphabet = "ACGTabcdefghijklm"; break;                     /* singleLevelYeast           ALL_YEAST            */
        case 19: alphabet = "ACGTabcdefghijklmnopq"; break;                 /* singleLevelYeastAltC       ALL_YEAST_ALTC       */
        case 20: alphabet = "ACGTabc"; break;                               /* singleLevelYeastSmall5mer  ALL_YEAST_SMALL_5MER */
        default:                                                            /* an unspecified type: a flat model over -b <alphabet> (:1403-1412) */
            if (!o->alphabet) die("loadNanoporeHdpFromScratch: an unspecified NanoporeHdpType needs an alphabet (-b)");
            alphabet = o->alphabet;
            break;
    }
    const int three = layout != SA_HDP_LAYOUT_FLAT;
    double g[3], ga[3], gb[3];
    if (!prior) {
        if (o->base_gamma == NULL_HYPERPARAMETER || o->leaf_gamma == NULL_HYPERPARAMETER || (three && o->middle_gamma == NULL_HYPERPARAMETER))
            die("loadNanoporeHdpFromScratch: You need to provide a base gamma, (middle gamma,) and leaf gamma for this NanoporeHdpType");
        g[0] = o->base_gamma;
        if (three) { g[1] = o->middle_gamma; g[2] = o->leaf_gamma; } else g[1] = o->leaf_gamma;
    } else {
        if (o->bga == NULL_HYPERPARAMETER || o->bgb == NULL_HYPERPARAMETER || o->lga == NULL_HYPERPARAMETER || o->lgb == NULL_HYPERPARAMETER ||
            (three && (o->mga == NULL_HYPERPARAMETER || o->mgb == NULL_HYPERPARAMETER)))
            die("loadNanoporeHdpFromScratch: You need to provide a alphas and betas for the base, (middle,) and the leaf distributions "
                "for the prior for this NanoporeHdp");
        ga[0] = o->bga; gb[0] = o->bgb;
        if (three) { ga[1] = o->mga; gb[1] = o->mgb; ga[2] = o->lga; gb[2] = o->lgb; } else { ga[1] = o->lga; gb[1] = o->lgb; }
    }
    sa_hdp_state_t *s = NULL;
    const int rc = sa_hdp_state_new(&s, layout, alphabet, o->kmer_length, gp, prior ? NULL : g, prior ? ga : NULL, prior ? gb : NULL,
                                    o->grid_start, o->grid_end, o->grid_length, mu, nu, alpha, beta);
    if (rc != SA_OK) {
        fprintf(stderr, "buildHdpUtil: cannot build the HDP (%s): check the k-mer length, the sampling grid and the gammas\n", sa_strerror(rc));
        exit(1);
    }
    return s;
}

static void sample_and_write(sa_hdp_state_t *s, const opts_t *o, const char *out_path) {
    int rc = sa_hdp_state_gibbs(s, o->n_samples, o->burn_in, o->thinning, o->seed, o->device, o->verbose);
    if (rc == SA_OK) rc = sa_hdp_state_finalize(s, o->device);
    if (rc != SA_OK) {
        fprintf(stderr, "buildHdpUtil: Gibbs sampling failed: %s\n", sa_strerror(rc));
        exit(1);
    }
    rc = sa_hdp_state_write(s, out_path);
    if (rc != SA_OK) {
        fprintf(stderr, "buildHdpUtil: cannot write %s\n", out_path);
        exit(1);
    }
}

static void build_strand(const opts_t *o, const char *what, const char *model_file, const char *alignments, const char *filter,
                         const char *out_path) {
    fprintf(stderr, "Updating %s HDP from alignments...\n", what);
    sa_hdp_state_t *s = from_scratch(o, model_file);
    int64_t n = 0;
    const int rc = sa_hdp_state_pass_assignment_file(s, alignments, filter, &n);
    if (rc == SA_EIO) {
        fprintf(stderr, "Alignment %s file does not exist or is not an assignments / alignment table.\n", alignments);
        exit(1);
    }
    if (rc != SA_OK) {
        fprintf(stderr, "buildHdpUtil: cannot take the assignments of strand %s from %s: %s\n", filter, alignments, sa_strerror(rc));
        exit(1);
    }
    fprintf(stderr, "Running Gibbs for %s doing %" PRId64 "samples, %" PRId64 "burn in, %" PRId64 "thinning.\n", what, o->n_samples, o->burn_in,
            o->thinning);
    sample_and_write(s, o, out_path);
    fprintf(stderr, "Serializing %s to %s...\n", what, out_path);
    sa_hdp_state_free(s);
}

int main(int argc, char *argv[]) {
    opts_t o;
    memset(&o, 0, sizeof(o));
    o.type = -1;
    o.base_gamma = o.middle_gamma = o.leaf_gamma = o.bga = o.bgb = o.mga = o.mgb = o.lga = o.lgb = NULL_HYPERPARAMETER;
    o.seed = 1;
    const char *t_table = NULL, *c_table = NULL, *alignments = NULL, *t_out = NULL, *c_out = NULL, *update_from = NULL, *expectations = NULL;
    int two_d = 1;
    static struct option long_options[] = {
        {"help", no_argument, 0, 'h'},
        {"verbose", no_argument, 0, 'o'},
        {"oneD", no_argument, 0, 'q'},
        {"kmerLength", required_argument, 0, 'a'},
        {"HdpType", required_argument, 0, 'p'},
        {"templateLookupTable", required_argument, 0, 'T'},
        {"complementLookupTable", required_argument, 0, 'C'},
        {"alignments", required_argument, 0, 'l'},
        {"templateHdp", required_argument, 0, 'v'},
        {"complementHdp", required_argument, 0, 'w'},
        {"nbSamples", required_argument, 0, 'n'},
        {"burnIn", required_argument, 0, 'I'},
        {"thinning", required_argument, 0, 't'},
        {"baseGamma", required_argument, 0, 'B'},
        {"middleGamma", required_argument, 0, 'M'},
        {"leafGamma", required_argument, 0, 'L'},
        {"baseGammaAlpha", required_argument, 0, 'g'},
        {"baseGammaBeta", required_argument, 0, 'r'},
        {"middleGammaAlpha", required_argument, 0, 'j'},
        {"middleGammaBeta", required_argument, 0, 'y'},
        {"leafGammaAlpha", required_argument, 0, 'i'},
        {"leafGammaBeta", required_argument, 0, 'u'},
        {"samplingGridStart", required_argument, 0, 's'},
        {"samplingGridEnd", required_argument, 0, 'e'},
        {"samplingGridLength", required_argument, 0, 'k'},
        {"alphabet", optional_argument, 0, 'b'},
        {"seed", required_argument, 0, 1001},
        {"device", required_argument, 0, 1002},
        {"updateFrom", required_argument, 0, 1003},
        {"expectations", required_argument, 0, 1004},
        {0, 0, 0, 0}};
    for (;;) {
        int option_index = 0;
        const int key = getopt_long(argc, argv, "h:a:o:q:p:T:C:l:v:w:n:I:t:B:M:L:g:r:j:b:y:i:u:s:e:k:", long_options, &option_index);
        if (key == -1) break;
        switch (key) {
            case 'h': usage(); return 1;
            case 'a': o.kmer_length = strtoll(optarg, NULL, 10); break;
            case 'p': o.type = strtoll(optarg, NULL, 10); break;
            case 'T': t_table = optarg; break;
            case 'C': c_table = optarg; break;
            case 'l': alignments = optarg; break;
            case 'v': t_out = optarg; break;
            case 'w': c_out = optarg; break;
            case 'b': o.alphabet = optarg; break;
            case 'n': o.n_samples = strtoll(optarg, NULL, 10); break;
            case 'I': o.burn_in = strtoll(optarg, NULL, 10); break;
            case 't': o.thinning = strtoll(optarg, NULL, 10); break;
            case 'q': two_d = 0; break;
            case 'o': o.verbose = 1; break;
            case 'B': o.base_gamma = strtod(optarg, NULL); break;
            case 'M': o.middle_gamma = strtod(optarg, NULL); break;
            case 'L': o.leaf_gamma = strtod(optarg, NULL); break;
            case 'g': o.bga = strtod(optarg, NULL); break;
            case 'r': o.bgb = strtod(optarg, NULL); break;
            case 'j': o.mga = strtod(optarg, NULL); break;
            case 'y': o.mgb = strtod(optarg, NULL); break;
            case 'i': o.lga = strtod(optarg, NULL); break;
            case 'u': o.lgb = strtod(optarg, NULL); break;
            case 's': o.grid_start = strtod(optarg, NULL); break;
            case 'e': o.grid_end = strtod(optarg, NULL); break;
            case 'k': o.grid_length = strtoll(optarg, NULL, 10); break;
            case 1001: o.seed = strtoull(optarg, NULL, 10); break;
            case 1002: o.device = atoi(optarg); break;
            case 1003: update_from = optarg; break;
            case 1004: expectations = optarg; break;
            default: usage(); return 1;
        }
    }
    if (o.n_samples < 1 || o.burn_in < 0 || o.thinning < 1) die("[buildHdpUtil] ERROR: need -n >= 1, -I >= 0 and -t >= 1");
    if (update_from) {
        /* updateHdpFromAssignments: deserialize_nhdp, hdpHmm_loadFromFile (the assignments go into the HDP), Gibbs, finalise, serialise */
        if (!expectations || !t_out) die("[buildHdpUtil] ERROR: --updateFrom needs --expectations <file> and -v <out.nhdp>");
        sa_hdp_state_t *s = NULL;
        sa_hmm_t *h = NULL;
        if (sa_hdp_state_load(&s, update_from) != SA_OK) die("[buildHdpUtil] ERROR: cannot read the .nhdp to update");
        if (sa_hmm_load(&h, expectations, SA_HMM_HDP, 0.0, 0.0) != SA_OK) die("ERROR loading hdpHmm");
        sa_hmm_view_t v;
        sa_hmm_view(h, &v);
        if (v.n_assignments < 1) die("[buildHdpUtil] ERROR: the expectations file holds no assignments");
        const int rc = sa_hdp_state_pass_assignments(s, v.assignment_kmers, v.assignment_events, v.n_assignments);
        if (rc != SA_OK) {
            fprintf(stderr, "[buildHdpUtil] ERROR: cannot pass the assignments to the HDP: %s\n", sa_strerror(rc));
            return 1;
        }
        sa_hmm_destroy(h);
        fprintf(stderr, "signalAlign - Running Gibbs on HDP doing %" PRId64 " samples %" PRId64 "burn in %" PRId64 "thinning\n", o.n_samples,
                o.burn_in, o.thinning);
        sample_and_write(s, &o, t_out);
        fprintf(stderr, "signalAlign - Serializing HDP to %s\n", t_out);
        sa_hdp_state_free(s);
        return 0;
    }
    if (t_out == NULL || (c_out == NULL && two_d)) die("[buildHdpUtil] ERROR: Need to specify where to put the HDP files");
    if (t_table == NULL || (c_table == NULL && two_d)) die("[buildHdpUtil] ERROR: Need lookup tables");
    /* printStartMessage */
    fprintf(stderr, "Building Nanopore HDP\n");
    fprintf(stderr, "Making HDP type %" PRId64 "\n", o.type);
    if (alignments != NULL) fprintf(stderr, "Using alignment from %s\n", alignments);
    fprintf(stderr, "Putting template here: %s\n", t_out);
    if (two_d) fprintf(stderr, "Putting complement here: %s\n", c_out);
    if (alignments == NULL) die("[buildHdpUtil]Need to provide build alignment (assignments)");
    if (o.kmer_length < 1) die("[buildHdpUtil] ERROR: need the k-mer length (-a)");
    fprintf(stderr, "Building Nanopore HDP\n");
    build_strand(&o, "template", t_table, alignments, "t", t_out);
    if (two_d) build_strand(&o, "complement", c_table, alignments, "c", c_out);
    return 0;
}
