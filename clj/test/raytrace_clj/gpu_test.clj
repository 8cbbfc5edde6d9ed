(ns raytrace-clj.gpu-test
  "The first test a maintainer of raytrace-clj runs after dropping clj/src/raytrace_clj/gpu.clj next to the reference's sources:

     lein test raytrace-clj.gpu-test          ; no GPU, no librtmi.so needed for `flatten-*`; `render-*` needs both

   flatten-*: the flattener of gpu.clj, run over the reference's OWN scene functions (scene.clj:9-48, 230-316, 318-412) with
   clojure.core/rand and rand-int rebound to a seeded sequence, must produce the flat arrays the MI355X build's Python mirror produced for the
   same scenes -- clj/test/resources/*.edn, exported from tests/golden/*.npz by scripts/export_clj_fixtures.py.  Those arrays are what the device and the
   CPU oracle render bit for bit, so a green flatten-* pins the Clojure host to the tested path: same primitives in the same (bvh-node left-to-right) order,
   same interned materials and textures, same camera.  It is also the harness of SURVEY.md's K16: with `rand` bound to the render stream instead
   (gpu/draw-bits of a sample key) the reference's own `pixel` can be diffed against rtmi_render.

   NEVER RUN where it was written (the build container has no JVM): tests/test_clj_conformance.py reads this file statically -- namespaces, the scene
   functions and gpu.clj vars it refers to, the fixture keys it reads."
  (:require [clojure.test :refer :all]
            [clojure.edn :as edn]
            [clojure.java.io :as io]
            [raytrace-clj.scene :as scene]
            [raytrace-clj.gpu :as gpu]))

;;; ---------------------------------------------------------------------------------------------
;;; the seeded sequence: splitmix64, exactly raytrace_clj_amd/util.py's SplitMix64 (scene construction only;
;;; the RENDER stream has its own mixer, include/rtmi.h)
;;; ---------------------------------------------------------------------------------------------

(def ^:private gold (unchecked-long 0x9E3779B97F4A7C15))

(defn- mix64 ^long [^long z]
  (let [z (bit-xor z (unsigned-bit-shift-right z 30))
        z (unchecked-multiply z (unchecked-long 0xBF58476D1CE4E5B9))
        z (bit-xor z (unsigned-bit-shift-right z 27))
        z (unchecked-multiply z (unchecked-long 0x94D049BB133111EB))]
    (bit-xor z (unsigned-bit-shift-right z 31))))

(defn- make-stream
  "replacements for clojure.core/rand and rand-int.
   rand     = (z >>> 11) * 2^-53 of a splitmix64 stream (the mirror's SplitMix64.rand; (rand n) = (* n (rand))): the scene's own numbers.
   rand-int = the next element of `axes`: the split axes make-bvh draws ((rand-int 3), hitable.clj:109), recorded in call order while the mirror built the
              fixture's scene.  Two sequences, because the ORDER between the two kinds of draws differs: the reference hands make-bvh a lazy item list
              (scene.clj:332: (make-bvh (concat (list ...) (for ...)))), so the outer tree's axis is drawn BEFORE the spheres' numbers, the mirror draws it
              after; among themselves the axes come in the same order (outer tree, then its halves depth first)."
  [seed axes]
  (let [state (atom (unchecked-long seed))
        left  (atom (seq axes))
        next! (fn ^long [] (mix64 (swap! state #(unchecked-add (long %) (long gold)))))
        unit  (fn [] (* (double (unsigned-bit-shift-right (next!) 11)) (/ 1.0 9007199254740992.0)))]
    {:rand     (fn ([] (unit)) ([n] (* n (unit))))
     :rand-int (fn [n]
                 (assert (= n 3) "only make-bvh draws integers in these scenes")
                 (let [a (first @left)]
                   (assert (some? a) "make-bvh drew more axes than the fixture recorded")
                   (swap! left next)
                   (int a)))
     :axes-left left}))

(defmacro with-seeded-rand
  "clojure.core is direct-linked since 1.8: core's own rand-int calls rand without going through the var, so BOTH vars are rebound.
   (The reference's namespaces are not direct-linked -- project.clj sets no :direct-linking -- and reach rand / rand-int through their vars.)
   The body must realise everything lazy before it returns: gpu/flatten-scene walks the whole world."
  [fx & body]
  `(let [s# (make-stream (:scene-seed ~fx) (:bvh-axes ~fx))
         out# (with-redefs [clojure.core/rand (:rand s#) clojure.core/rand-int (:rand-int s#)]
                ~@body)]
     (is (empty? @(:axes-left s#)) "make-bvh drew every recorded axis")
     out#))

;;; ---------------------------------------------------------------------------------------------
;;; fixtures
;;; ---------------------------------------------------------------------------------------------

(defn- fixture [name]
  (edn/read-string {:readers {} :default (fn [_ v] v)} (slurp (io/resource (str name ".edn")))))

(defn- ulps-apart
  "0 when a and b are the same double; else how many representable doubles lie between them (same sign)"
  [^double a ^double b]
  (if (== a b) 0 (Math/abs (- (Double/doubleToLongBits a) (Double/doubleToLongBits b)))))

(defn- check-flat
  "f: what gpu/flatten-scene returned; fx: the fixture.  Integer tables and the scene's own numbers (centres, radii, colours: doubles the scene
   functions write down or draw from the seeded sequence) must agree exactly; the camera goes through vectorz (normalise, cross: camera.clj:56-66),
   whose last bit the mirror restates but the reference's tests do not pin (SURVEY.md 8c): two ulps."
  [f fx]
  (doseq [[k fk] [[:prim-kind :prim-kind] [:prim-mat :prim-mat] [:mat-kind :mat-kind] [:mat-tex :mat-tex] [:tex-kind :tex-kind] [:tex-child :tex-child]
                  [:prim-flip :prim-flip] [:prim-xform :prim-xform] [:xform-kind :xform-kind]]
          :when (contains? fx fk)]
    (is (= (vec (get f k)) (get fx fk)) (str k)))
  (doseq [[k fk] [[:prim-geom :prim-geom] [:mat-param :mat-param] [:tex-param :tex-param]]]
    (is (= (count (get f k)) (count (get fx fk))) (str k " length"))
    (is (every? zero? (map ulps-apart (get f k) (get fx fk))) (str k)))
  (when (contains? fx :xform-param) ; RotateY's sin / cos go through Math/sin, Math/cos of (* theta (/ Math/PI 180)) (hitable.clj:466-470): one ulp
    (is (every? #(<= % 1) (map ulps-apart (:xform-param f) (:xform-param fx))) ":xform-param"))
  (is (= (:cam-kind f) (:cam-kind fx)))
  (is (every? #(<= % 2) (map ulps-apart (:cam f) (:cam fx))) ":cam"))

(deftest flatten-two-spheres
  (let [fx (fixture "two_spheres")]
    (check-flat (with-seeded-rand fx (gpu/flatten-scene (scene/make-two-spheres (:nx fx) (:ny fx)))) fx)))

(deftest flatten-cornell-box
  (let [fx (fixture "cornell_box")]
    (check-flat (with-seeded-rand fx (gpu/flatten-scene (scene/make-cornell-box (:nx fx) (:ny fx)))) fx)))

(deftest flatten-cover-scene
  ;; scene.clj:318-412 with n = 3, static spheres: 36 lattice cells, each drawing centre x, centre z, choose-mat and the material's numbers in that order
  (let [fx (fixture "cover_n3")]
    (check-flat (with-seeded-rand fx (gpu/flatten-scene (scene/make-random-scene (:nx fx) (:ny fx) 3 false))) fx)))

(deftest flatten-is-order-stable
  ;; a one-item make-bvh stores its item twice (hitable.clj:113-114): the flattener drops the repeat, so flattening twice gives the same arrays
  (let [fx (fixture "two_spheres")
        f1 (with-seeded-rand fx (gpu/flatten-scene (scene/make-two-spheres (:nx fx) (:ny fx))))
        f2 (with-seeded-rand fx (gpu/flatten-scene (scene/make-two-spheres (:nx fx) (:ny fx))))]
    (is (= (vec (:prim-geom f1)) (vec (:prim-geom f2))))
    (is (= (:n-prims f1) (count (:prim-kind fx))))))

;;; ---------------------------------------------------------------------------------------------
;;; with an MI355X and librtmi.so on jna.library.path: the frame of the flattened fixture scene
;;; ---------------------------------------------------------------------------------------------

(deftest ^:gpu render-two-spheres
  ;; gpu/render returns {:rgb8 bytes :linear doubles :total-rays n :total-pixels n}; the 8-bit frame of tests/golden/render_two_spheres.npz has
  ;; 40 x 20 x 3 bytes and 800 pixels
  (let [fx  (fixture "two_spheres")
        out (with-seeded-rand fx (gpu/render (scene/make-two-spheres (:nx fx) (:ny fx)) (:nx fx) (:ny fx) 4))]
    (is (= (* 40 20 3) (count (:rgb8 out))))
    (is (= 800 (:total-pixels out)))
    (is (pos? (:total-rays out)))))
