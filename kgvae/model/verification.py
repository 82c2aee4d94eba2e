"""Semantic verification hooks (reference kgvae/model/verification.py).  The rule checkers live in
the third-party `intelligraphs` package; when it is not installed these helpers return None and the
training loop skips the periodic verification, exactly as it does for a dataset without a
verifier (reference train.py:418-421, 513)."""


def get_verifier(dataset_name):
    try:
        from intelligraphs.verifier.synthetic import SynPathsVerifier, SynTIPRVerifier, SynTypesVerifier
        from intelligraphs.verifier.wikidata import WDArticlesVerifier, WDMoviesVerifier
    except Exception:
        return None
    table = {"syn-paths": SynPathsVerifier, "syn-tipr": SynTIPRVerifier, "syn-types": SynTypesVerifier,
             "wd-movies": WDMoviesVerifier, "wd-articles": WDArticlesVerifier}
    cls = table.get(dataset_name)
    return cls() if cls is not None else None


def run_semantic_evaluation(label_graphs, train_graphs, i2e, i2r, verifier, title="samples"):
    """validity / novelty of generated graphs through intelligraphs' SemanticEvaluator"""
    from intelligraphs.evaluators import SemanticEvaluator, post_process_data
    from kgvae.model.utils import ints_to_labels
    train_labels = ints_to_labels(train_graphs, i2e, i2r)
    ev = SemanticEvaluator(post_process_data(label_graphs), post_process_data(train_labels), verifier.check_rules_for_graph,
                           entity_labels=list(i2e.values()), relation_labels=list(i2r.values()))
    ev.evaluate_graphs()
    print(f"[{title}]")
    ev.print_results()
    return ev
