"""CPU oracle for the ESRecsys hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the arithmetic the reference
(BBischof/ESRecsys) executes on its training hot path:

  * wikipedia/models.py:15-55          Glove.setup / __call__ / score_all
  * wikipedia/train_cooccurence.py:71-101   apply_model / find_knn / update_model
  * pinterest/models.py:63-74          STLModel score head
  * pinterest/train_shop_the_look.py:93-122  train_step / eval_step
  * pinterest/make_recommendations.py:49-65  find_top_k

PARITY UNPINNED.  The reference has no tests, golden vectors or fixtures for
this path and its arithmetic lives in un-vendored
third-party packages (jax==0.3.25, jaxlib==0.3.22, flax==0.5.2,
optax==0.1.2 -- wikipedia/requirements.txt:18-21, pinterest/requirements.txt:5-8)
that are not installed here, so the reference cannot be imported or run.  The
oracle is therefore pinned only by (a) the source text of the functions cited
above, restated line by line, and (b) an independent torch-CPU-autograd
transliteration of the same jnp expressions (oracle/autograd_ref.py) that
must agree with the closed-form gradients here to <= 1e-12 in fp64
(tests/test_oracle.py).  Upstream semantics that are restated from the pinned
versions' published behaviour (optax.adam / optax.adagrad update rules,
jax.nn.relu'(0) = 0, jax.lax.top_k tie order, stable jnp.argsort) are flagged
"[upstream]" where they are used.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  Nothing under esrecsys_amd/ imports it; the product path
has no CPU fallback.
"""
from . import glove, stl_head, optim, topk, shard  # noqa: F401
