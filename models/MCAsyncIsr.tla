---------------------------- MODULE MCAsyncIsr ----------------------------
(* Model-checking wrapper for AsyncIsr.tla of hachikuji/kafka-specification.  AUTHORED HERE — the
   reference repository contains no such module.

   AsyncIsr is unbounded as written: `version: Nat` and `offsets: [Replicas -> Nat]`
   (AsyncIsr.tla:40-56), LeaderWrite (:117-119) is always enabled, and the controller can bump
   its version for ever by re-admitting a replica it has just removed.  MaxOffset (:25) only
   feeds the definition `Offsets` (:37), which nothing uses.  TLC therefore needs a state
   constraint to terminate; this module supplies the smallest one that bounds both counters. *)
EXTENDS AsyncIsr

CONSTANT MaxVersion
ASSUME MaxVersion \in Nat

StateConstraint ==
    /\ leaderState.offsets[Leader] <= MaxOffset
    /\ controllerState.version <= MaxVersion

(* What the otherwise unused `Offsets` suggests TypeOk was meant to say about the leader's log
   end.  Under StateConstraint it is false exactly in the successors of LeaderWrite that leave the
   constraint — states TLC still checks invariants on although it does not explore them. *)
LeaderOffsetInRange == leaderState.offsets[Leader] \in Offsets
=============================================================================
