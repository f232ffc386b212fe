"""Drop-in module paths of the reference package: the Monte-Carlo scripts import
``DGSQP.solvers.DGSQP``, ``DGSQP.solvers.solver_types``, ``DGSQP.types``, ``DGSQP.dynamics.dynamics_models``,
``DGSQP.dynamics.model_types``, ``DGSQP.tracks.track_lib`` (scripts/DGSQP_ALGAMES_monte_carlo_curve.py:5-15).
Every module here only re-exports the names of ``dgsqp_amd`` under the reference's path; nothing is implemented here."""
