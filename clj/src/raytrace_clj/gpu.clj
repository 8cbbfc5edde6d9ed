(ns raytrace-clj.gpu
  "GPU path for the render loop of raytrace-clj.core/-main (core.clj:100-108 of the reference).

  A scene built at the REPL from the reference's own records -- {:camera c :world w} as every
  make-* function of raytrace-clj.scene returns -- is flattened into the primitive arrays of
  include/rtmi.h and rendered by librtmi.so (HIP kernels for MI355X) through JNA.  Nothing is
  computed on the JVM; unknown record types raise ex-info {:unsupported-on-gpu-path ...} so the
  caller can fall back to the protocol path (raytrace-clj.core/pixel).

  STATUS: written blind (no JVM/lein/JNA jar in the build container); the Python mirror in
  raytrace_clj_amd/ exercises the same C-ABI call for call and is what the tests run.  What CAN be checked without a JVM is:
  tests/test_clj_conformance.py reads this file's (call-int \"rtmi_...\" ...) forms against the prototypes of include/rtmi.h --
  symbol, arity, (int ...) / (long ...) coercion of every scalar, element type of every array, handle vs handle-by-reference."
  (:require [clojure.core.matrix :as mat]
            [raytrace-clj.scene :as scene]
            [raytrace-clj.perlin :as perlin]
            [mikera.image.core :refer [new-image set-pixel save]]
            [mikera.image.colours :refer [rgb-from-components]])
  (:import [com.sun.jna Function Pointer Memory]
           [com.sun.jna.ptr PointerByReference]
           [raytrace_clj.hitable Hitlist bvh_node Sphere UVSphere MovingSphere RectXY RectXZ RectYZ Triangle
            FlipNormals Translate RotateY Box ConstantMedium]
           [raytrace_clj.shader Lambertian Metal Dielectric DiffuseLight Isotropic]
           [raytrace_clj.texture Constant UVGradient Checkerboard PerlinNoise PerlinTurbulence Marble
            FlipTextureU FlipTextureV ImageMap]
           [raytrace_clj.camera PinholeCamera ThinLensCamera]))

(mat/set-current-implementation :vectorz)

;;; ---------------------------------------------------------------------------------------------
;;; JNA plumbing: plain C functions, int return codes, rtmi_last_error for the text
;;; ---------------------------------------------------------------------------------------------

(def ^:private lib "rtmi")

(defn- cfn ^Function [name] (Function/getFunction lib name))

(defn- check [rc]
  (when-not (zero? rc)
    (throw (ex-info (str "rtmi error " rc ": "
                         (.invokeString (cfn "rtmi_last_error") (object-array 0) false))
                    {:rtmi-code rc}))))

(defn- call-int [name & args]
  (.invokeInt (cfn name) (object-array args)))

;;; ---------------------------------------------------------------------------------------------
;;; flattener: protocol extended onto the reference's records (field names as in the reference)
;;; ---------------------------------------------------------------------------------------------

(defn- v3 [v] [(mat/mget v 0) (mat/mget v 1) (mat/mget v 2)])

(def ^:private ^:dynamic *list-ctx* nil)   ; id of the Hitlist context being walked: the outermost Hitlist since the last bvh-node (nil outside any list)
(def ^:private ^:dynamic *slot* nil)       ; [identity of the list, position] of the list item being walked: a medium's LISTING
(def ^:private ^:dynamic *poisoned* false) ; a bvh-node INSIDE a Hitlist is among the ancestors: media below it cannot be expressed
(def ^:private list-ctxs (atom 0))         ; counts the Hitlist contexts of one flatten-scene walk
(def ^:private media-modes (atom #{}))   ; how flatten-scene's walk reached the media: :descent (bvh-nodes only above) / :hitlist (Hitlists only above)

(defprotocol GpuLeaves
  (leaves [this chain flip in-list? under-bvh?]
    "[{:leaf record :chain [[kind a b c] ...] :flip 0|1} ...] below this Hitable, in the order (and as often as) the
    reference's descent calls hit? on them; chain = the Translate / RotateY wrappers around the leaf, outermost first
    (kind 0 = translate offset.xyz, 1 = rotate-y sin cos 0).  A one-item make-bvh node holds the same child twice
    (hitable.clj:113-114) and is therefore listed twice: dedup-leaves drops the repeats for the primitive table, the
    repeats of ConstantMedium records are kept as the media call sequence.  in-list? / under-bvh?: a Hitlist / a bvh-node is among the
    ancestors -- a medium's hit? draws inside, so it matters whether its t-max comes narrowed by the siblings before it (Hitlist.hit?,
    hitable.clj:15-26: rtmi_scene_set_media_mode RTMI_MEDIA_HITLIST) or un-narrowed (bvh-node.hit?, hitable.clj:99-105); both at once is
    not supported."))

(defn- leaf [this chain flip] [{:leaf this :chain chain :flip flip :list *list-ctx*}])   ; :list -- which Hitlist context the entry stands in (its items are contiguous)

(extend-protocol GpuLeaves
  Hitlist      (leaves [this c f l b]                                                                      ; hitable.clj:15-26
                 ;; the items of a Hitlist (and of the Hitlists nested directly in it) see the closest hit of the items before them: one narrowing context
                 (binding [*list-ctx* (if (and l *list-ctx*) *list-ctx* (swap! list-ctxs inc))]
                   (doall (apply concat (map-indexed (fn [i it] (binding [*slot* [(System/identityHashCode this) i]] (doall (leaves it c f true b))))
                                                     (:items this))))))
  bvh_node     (leaves [this c f l b]                                                                      ; hitable.clj:97-105: both children, the un-narrowed interval
                 (binding [*list-ctx* nil *slot* nil *poisoned* (or *poisoned* (boolean l))]
                   (doall (concat (leaves (:left this) c f false true) (leaves (:right this) c f false true)))))
  Box          (leaves [this c f l b] (leaves (:sides this) c f l b))   ; hitable.clj:491-494: (hit? sides ...) with the caller's interval
  FlipNormals  (leaves [this c f l b] (leaves (:item this) c (bit-xor f 1) l b))                             ; hitable.clj:375-381
  Translate    (leaves [this c f l b] (leaves (:item this) (conj c (into [0.0] (v3 (:offset this)))) f l b)) ; hitable.clj:391-396
  RotateY      (leaves [this c f l b] (leaves (:obj this) (conj c [1.0 (:sin-theta this) (:cos-theta this) 0.0]) f l b)) ; :410-450
  ConstantMedium (leaves [this c f l b]                                                                      ; hitable.clj:516-541
                   (when *poisoned*
                     (throw (ex-info "a ConstantMedium below a bvh-node that itself sits inside a Hitlist is not supported on the GPU path"
                                     {:unsupported-on-gpu-path ConstantMedium})))
                   (swap! media-modes conj (cond (and l b) :narrowed l :hitlist :else :descent))
                   ;; inside a Hitlist WHERE a medium stands decides the t-max it is handed (hitable.clj:15-26): every LISTING of the record is a primitive of
                   ;; its own (:listing = the list item's slot: the same listing reached twice by the descent -- a one-item make-bvh -- is one primitive asked twice)
                   [{:leaf this :chain c :flip f :list *list-ctx* :listing (when l *slot*)}])
  Sphere       (leaves [this c f l b] (leaf this c f))
  UVSphere     (leaves [this c f l b] (leaf this c f))
  MovingSphere (leaves [this c f l b] (leaf this c f))
  RectXY       (leaves [this c f l b] (leaf this c f))
  RectXZ       (leaves [this c f l b] (leaf this c f))
  RectYZ       (leaves [this c f l b] (leaf this c f))
  Triangle     (leaves [this c f l b] (leaf this c f))
  Object       (leaves [this c f l b] (throw (ex-info (str (type this) " is not supported on the GPU path")
                                                      {:unsupported-on-gpu-path (type this)}))))

(defn- leaf-id-fn
  "object IDENTITY -> small integer (an IdentityHashMap, not System/identityHashCode: that is a 31-bit hash and two of the
  ~10 000 spheres of the cover scene at n = 50 collide with probability ~2 %)"
  []
  (let [ids (java.util.IdentityHashMap.)]
    (fn [leaf]
      (or (.get ids leaf)
          (let [k (.size ids)] (.put ids leaf k) k)))))

(defn- dedup-leaves
  "a one-item make-bvh stores the same child twice (hitable.clj:113-114): drop repeats of the same record under the
  same instance chain (the same record under two different instances is two primitives)"
  [lid xs]
  (let [seen (java.util.HashSet.)]
    (filterv #(.add seen [(lid (:leaf %)) (:chain %) (:flip %) (:listing %)]) xs)))

(defn- intern! [table obj build]
  ;; table: atom {:ids IdentityHashMap, :rows []}; children are interned first
  (let [^java.util.IdentityHashMap ids (:ids @table)]
    (or (.get ids obj)
        (let [row (build obj)
              id  (count (:rows @table))]
          (.put ids obj id)
          (swap! table update :rows conj row)
          id))))

(def ^:private images (atom []))   ; BufferedImages met while flattening, in ImageMap index order

(defn- pad12 [xs] (take 12 (concat (map double xs) (repeat 0.0))))

(defn- tex-row [textures t]
  (condp instance? t
    Constant     {:kind 0 :param (concat (v3 (:color t)) (repeat 9 0.0)) :child [-1 -1]}
    UVGradient   {:kind 1 :param (mapcat v3 [(:co t) (:cu t) (:cv t) (:cuv t)]) :child [-1 -1]}
    PerlinNoise      {:kind 3 :param (pad12 [(:scale t)]) :child [-1 -1]}                              ; texture.clj:60
    PerlinTurbulence {:kind 4 :param (pad12 [(:scale t) (:depth t)]) :child [-1 -1]}                   ; texture.clj:74
    Marble           {:kind 5 :param (pad12 [(:scale t) (:depth t)]) :child [-1 -1]}                   ; texture.clj:88
    FlipTextureU     {:kind 6 :param (pad12 []) :child [(intern! textures (:tex t) (partial tex-row textures)) -1]} ; :103
    FlipTextureV     {:kind 7 :param (pad12 []) :child [(intern! textures (:tex t) (partial tex-row textures)) -1]} ; :113
    ImageMap         (let [id (count @images)]                                                         ; texture.clj:126
                       (swap! images conj (:image t))
                       {:kind 8 :param (pad12 [id]) :child [-1 -1]})
    Checkerboard (let [c0 (intern! textures (:tex0 t) (partial tex-row textures))
                       c1 (intern! textures (:tex1 t) (partial tex-row textures))]
                   {:kind 2 :param (cons (double (:scale t)) (repeat 11 0.0)) :child [c0 c1]})
    (throw (ex-info (str (type t) " is not supported on the GPU path") {:unsupported-on-gpu-path (type t)}))))

(defn- mat-row [textures m]
  (let [tex (fn [t] (intern! textures t (partial tex-row textures)))]
    (condp instance? m
      Lambertian   {:kind 0 :tex (tex (:albedo m)) :param 0.0}
      Metal        {:kind 1 :tex (tex (:albedo m)) :param (double (:fuzz m))}
      Dielectric   {:kind 2 :tex -1 :param (double (:ri m))}
      DiffuseLight {:kind 3 :tex (tex (:tex m)) :param 0.0}
      Isotropic    {:kind 4 :tex (tex (:albedo m)) :param 0.0}                    ; shader.clj:129, a medium's phase function
      (throw (ex-info (str (type m) " is not supported on the GPU path") {:unsupported-on-gpu-path (type m)})))))

(defn- pad9 [xs] (take 9 (concat (map double xs) (repeat 0.0))))

(defn- prim-row [o]
  (condp instance? o
    MovingSphere {:kind 2 :geom (concat (v3 (:center0 o)) [(double (:radius o))] (v3 (:center1 o))
                                        [(double (:t0 o)) (double (:t1 o))])}
    UVSphere     {:kind 1 :geom (concat (v3 (:center o)) [(double (:radius o))] (v3 (:center o)) [0.0 1.0])}
    Sphere       {:kind 0 :geom (concat (v3 (:center o)) [(double (:radius o))] (v3 (:center o)) [0.0 1.0])}
    RectXY       {:kind 3 :geom (pad9 [(:x0 o) (:y0 o) (:x1 o) (:y1 o) (:k o)])}   ; hitable.clj:269
    RectXZ       {:kind 4 :geom (pad9 [(:x0 o) (:z0 o) (:x1 o) (:z1 o) (:k o)])}   ; hitable.clj:301
    RectYZ       {:kind 5 :geom (pad9 [(:y0 o) (:z0 o) (:y1 o) (:z1 o) (:k o)])}   ; hitable.clj:333
    Triangle     {:kind 6 :geom (concat (v3 (:v0 o)) (v3 (:v1 o)) (v3 (:v2 o)))}   ; hitable.clj:548
    ConstantMedium {:kind 7 :geom (pad9 [(:density o)])}))  ; hitable.clj:516; boundary range patched in by flatten-scene

(defn- camera-row [c]
  (condp instance? c
    ThinLensCamera {:kind 1 :cam (concat (mapcat v3 [(:origin c) (:lleft c) (:horiz c) (:vert c) (:u c) (:v c) (:w c)])
                                         [(double (:aperture c)) (double (:t0 c)) (double (:t1 c))])}
    PinholeCamera  {:kind 0 :cam (concat (mapcat v3 [(:origin c) (:lleft c) (:horiz c) (:vert c)]) (repeat 12 0.0))}
    (throw (ex-info (str (type c) " is not supported on the GPU path") {:unsupported-on-gpu-path (type c)}))))

(defn flatten-scene
  "{:camera c :world w} -> the flat arrays of include/rtmi.h (as Clojure primitive arrays)"
  [{:keys [camera world]}]
  (reset! images [])
  (reset! list-ctxs 0)
  (reset! media-modes #{})
  (let [called    (vec (leaves world [] 0 false false))    ; hit? invocation order, repeats included
        _         (when (> (count @media-modes) 1)
                    (throw (ex-info "media reached through a Hitlist and media reached through bvh-nodes only in one world: not supported on the GPU path"
                                    {:unsupported-on-gpu-path ConstantMedium})))
        lid       (leaf-id-fn)
        world-es  (dedup-leaves lid called)
        key-of    (fn [e] [(lid (:leaf e)) (:chain e) (:flip e) (:listing e)])
        index-of  (zipmap (map key-of world-es) (range))
        called-media (filterv #(instance? ConstantMedium (:leaf %)) called)
        media-calls (mapv #(index-of (key-of %)) called-media)
        ;; per call: the first primitive of the Hitlist items that narrow it (rtmi_scene_set_media_calls_narrowed); its own index: none (reached through bvh-nodes only)
        first-of  (reduce (fn [m [i e]] (if (and (:list e) (not (contains? m (:list e)))) (assoc m (:list e) i) m)) {} (map-indexed vector world-es))
        media-narrow-from (mapv (fn [e] (if (:listing e) (first-of (:list e)) (index-of (key-of e)))) called-media)
        ;; every medium's boundary is flattened on its own and appended AFTER the world (kind | 16)
        bounds    (reduce (fn [acc [i e]]
                            (if (instance? ConstantMedium (:leaf e))
                              (let [b (dedup-leaves (leaf-id-fn) (leaves (:boundary (:leaf e)) (:chain e) (:flip e) false false))]
                                (-> acc
                                    (assoc-in [:range i] [(+ (count world-es) (count (:prims acc))) (count b)])
                                    (update :prims into b)))
                              acc))
                          {:range {} :prims []}
                          (map-indexed vector world-es))
        n-world   (count world-es)
        entries   (into (vec world-es) (:prims bounds))
        prims     (mapv :leaf entries)
        material-of (fn [o] (if (instance? ConstantMedium o) (:phase-fn o) (:material o)))
        textures  (atom {:ids (java.util.IdentityHashMap.) :rows []})
        materials (atom {:ids (java.util.IdentityHashMap.) :rows []})
        prim-mat  (mapv #(intern! materials (material-of %) (partial mat-row textures)) prims)
        prows     (vec (map-indexed
                        (fn [i o]
                          (let [row (prim-row o)]
                            (cond
                              (instance? ConstantMedium o)
                              (let [[fb nb] (get-in bounds [:range i])]
                                (assoc row :geom (pad9 [(:density o) fb nb])))
                              (>= i n-world) (update row :kind bit-or 16)
                              :else row)))
                        prims))
        ;; transform table: every distinct chain once; per primitive [first count]
        chains    (vec (distinct (remove empty? (map :chain entries))))
        starts    (reductions + 0 (map count chains))
        chain-at  (zipmap chains starts)
        xforms    (vec (mapcat identity chains))
        mrows     (:rows @materials)
        trows     (:rows @textures)
        cam       (camera-row camera)]
    {:n-prims   (count prims)
     :prim-kind (int-array (map :kind prows))
     :prim-geom (double-array (mapcat :geom prows))
     :prim-mat  (int-array prim-mat)
     :n-mats    (count mrows)
     :mat-kind  (int-array (map :kind mrows))
     :mat-tex   (int-array (map :tex mrows))
     :mat-param (double-array (map :param mrows))
     :n-tex     (count trows)
     :tex-kind  (int-array (map :kind trows))
     :tex-param (double-array (mapcat :param trows))
     :tex-child (int-array (mapcat :child trows))
     :cam-kind  (int (:kind cam))
     :cam       (double-array (:cam cam))
     :prim-flip  (int-array (map :flip entries))
     :prim-xform (int-array (mapcat (fn [e] (if (empty? (:chain e)) [0 0] [(chain-at (:chain e)) (count (:chain e))])) entries))
     :n-xforms   (count xforms)
     :xform-kind  (int-array (map #(int (first %)) xforms))
     :xform-param (double-array (mapcat rest xforms))
     :media-calls (int-array media-calls)
     :media-narrow-from (int-array media-narrow-from)
     ;; 1 = RTMI_MEDIA_HITLIST: the world is a Hitlist, its narrowing reaches every medium; 2 = narrowed per call: Hitlists holding media below bvh-nodes
     :media-mode  (int (cond (= @media-modes #{:hitlist}) 1 (or (:narrowed @media-modes) (:hitlist @media-modes)) 2 :else 0))
     :uses-perlin (boolean (some #(#{3 4 5} (:kind %)) trows))
     :images      @images}))

;;; ---------------------------------------------------------------------------------------------
;;; render: replaces (dorun (cp/upmap ...)) of core.clj:100-108
;;; ---------------------------------------------------------------------------------------------

(defn- create-scene!
  "rtmi_scene_create_ex + everything a scene may need before its first render -- the namespace-level Perlin tables of the running
  JVM (perlin.clj:6-17), the decoded ImageMap pixels (texture.clj:126-133), the media call sequence -- on context `ctx` (a Pointer).
  Returns the scene Pointer.  Clones (rtmi_scene_clone) carry all of it."
  [ctx f]
  (let [scn (PointerByReference.)]
    (check (call-int "rtmi_scene_create_ex" ctx
                     (int (:n-prims f)) (:prim-kind f) (:prim-geom f) (:prim-mat f)
                     (int (:n-mats f)) (:mat-kind f) (:mat-tex f) (:mat-param f)
                     (int (:n-tex f)) (:tex-kind f) (:tex-param f) (:tex-child f)
                     (int (:cam-kind f)) (:cam f)
                     (:prim-flip f) (:prim-xform f) (int (:n-xforms f)) (:xform-kind f) (:xform-param f) scn))
    (let [scene (.getValue scn)]
      (when (:uses-perlin f)
        (let [vectors (double-array (mapcat v3 perlin/random-vectors))
              perm    (int-array (concat perlin/perm-x perlin/perm-y perlin/perm-z))]
          (check (call-int "rtmi_scene_set_perlin" scene vectors perm))))
      ;; ImageMap pixels: rows top-down, RGB (imagez get-pixel = BufferedImage.getRGB, texture.clj:76)
      (when (seq (:images f))
        (let [imgs (:images f)
              wh   (int-array (mapcat (fn [^java.awt.image.BufferedImage im] [(.getWidth im) (.getHeight im)]) imgs))
              px   (byte-array (mapcat (fn [^java.awt.image.BufferedImage im]
                                         (for [y (range (.getHeight im)) x (range (.getWidth im))
                                               :let [p (.getRGB im (int x) (int y))]
                                               sh [16 8 0]]
                                           (unchecked-byte (bit-and 0xff (bit-shift-right p sh)))))
                                       imgs))]
          (check (call-int "rtmi_scene_set_images" scene (int (count imgs)) wh px))))
      (if (= 2 (:media-mode f))
        (let [calls (:media-calls f)]
          (check (call-int "rtmi_scene_set_media_calls_narrowed" scene (int (alength ^ints calls)) calls (:media-narrow-from f))))
        (do
          (when (pos? (alength ^ints (:media-calls f)))
            (let [calls (:media-calls f)]
              (check (call-int "rtmi_scene_set_media_calls" scene (int (alength ^ints calls)) calls))))
          (when (pos? (:media-mode f))
            (check (call-int "rtmi_scene_set_media_mode" scene (int (:media-mode f)))))))
      scene)))

(defn render
  "Render scene {:camera :world} at nx x ny with ns samples per pixel on GPU `device`.
  Returns {:rgb8 byte-array (row 0 = top, RGB interleaved) :linear double-array
           :total-rays n :total-pixels n} (the counters of metrics.clj:8-9)."
  [scene nx ny ns & {:keys [depth seed device precision] :or {depth 50 seed 0x5eed0002 device 0 precision 0}}]
  (let [f    (flatten-scene scene)
        ctx  (PointerByReference.)
        npx  (* nx ny)
        lin  (double-array (* 3 npx))
        rgb  (byte-array (* 3 npx))
        cnt  (long-array 2)]
    (check (call-int "rtmi_init" (int device) (int 0) ctx))
    (try
      (let [scn (create-scene! (.getValue ctx) f)]
        (try
          (check (call-int "rtmi_render" scn (int nx) (int ny) (int ns) (int depth) (long seed) (int precision)
                           (int 0) (int 0) (int nx) (int ny) lin rgb cnt))
          {:rgb8 rgb :linear lin :total-rays (aget cnt 0) :total-pixels (aget cnt 1)}
          (finally (call-int "rtmi_scene_destroy" scn))))
      (finally (call-int "rtmi_shutdown" (.getValue ctx))))))

(defn render-multi
  "The same on every GPU in `devices` from this one JVM (rtmi_render_multi): one context per device, the scene created on
  the first (Perlin tables, ImageMap pixels and media calls included) and cloned onto the others (rtmi_scene_clone), the 8x8
  tiles of the frame dealt round-robin, ONE ncclGather inside the library, assembled on the first device.  Returns what
  `render` returns; :total-rays is the sum over the devices.  The image is bit-identical to `render`'s."
  [scene nx ny ns & {:keys [depth seed devices precision] :or {depth 50 seed 0x5eed0002 devices [0] precision 0}}]
  (let [f     (flatten-scene scene)
        npx   (* nx ny)
        lin   (double-array (* 3 npx))
        rgb   (byte-array (* 3 npx))
        cnt   (long-array 2)
        ctxs  (mapv (fn [d] (let [ctx (PointerByReference.)] (check (call-int "rtmi_init" (int d) (int 0) ctx)) (.getValue ctx))) devices)]
    (try
      (let [scn0   (create-scene! (first ctxs) f)
            clones (mapv (fn [c] (let [scn (PointerByReference.)] (check (call-int "rtmi_scene_clone" scn0 c scn)) (.getValue scn)))
                         (rest ctxs))
            scenes (into [scn0] clones)
            handles (into-array com.sun.jna.Pointer scenes)]
        (try
          (check (call-int "rtmi_render_multi" (int (count scenes)) handles
                           (int nx) (int ny) (int ns) (int depth) (long seed) (int precision) lin rgb cnt))
          {:rgb8 rgb :linear lin :total-rays (aget cnt 0) :total-pixels (aget cnt 1)}
          (finally (doseq [s scenes] (call-int "rtmi_scene_destroy" s)))))
      (finally (doseq [c ctxs] (call-int "rtmi_shutdown" c))))))

(defn save-ppm
  "imagez `save` (core.clj:112) has no PPM writer; binary P6 written here"
  [filename ^bytes rgb8 nx ny]
  (with-open [o (java.io.FileOutputStream. ^String filename)]
    (.write o (.getBytes (format "P6\n%d %d\n255\n" nx ny)))
    (.write o rgb8)))

(defn save-image
  "what core.clj:104-106,112 does with the pixels: an imagez image saved by extension (PNG by default, core.clj:76).
  rgb8 is row-major with row 0 = top, so no flip is needed here."
  [filename ^bytes rgb8 nx ny]
  (let [image (new-image nx ny)]
    (doseq [j (range ny) i (range nx)]
      (let [o (* 3 (+ i (* j nx)))]
        (set-pixel image i j (rgb-from-components (bit-and 0xff (aget rgb8 o))
                                                  (bit-and 0xff (aget rgb8 (+ o 1)))
                                                  (bit-and 0xff (aget rgb8 (+ o 2)))))))
    (save image filename)))

(defn -main
  "lein run name nx ny ns -- same positional arguments as raytrace-clj.core/-main (core.clj:73-80);
  renders the cover scene (the commented-out line core.clj:89)."
  [& [name ix iy is which]]
  (let [tstart   (System/currentTimeMillis)
        filename (or name "render.png")
        nx (if ix (Integer/parseUnsignedInt ix) 200)
        ny (if iy (Integer/parseUnsignedInt iy) 100)
        nr (if is (Integer/parseUnsignedInt is) 100)
        sc (if (= which "final")
             (scene/make-final nx ny)                 ; core.clj:90 (needs earth.png in the working directory)
             (scene/make-random-scene nx ny 11 true)) ; core.clj:89
        {:keys [rgb8 total-rays total-pixels]} (render sc nx ny nr)
        elapsed (/ (- (System/currentTimeMillis) tstart) 1000.0)]
    (println (format "%.2fs, %d%%, ETA %.2fs" elapsed 100 0.0))      ; display.clj:20-24
    (println "total-rays" total-rays "total-pixels" total-pixels)    ; metrics.clj:8-9
    (if (.endsWith (.toLowerCase ^String filename) ".ppm")
      (save-ppm filename rgb8 nx ny)
      (save-image filename rgb8 nx ny))
    (println "wrote" filename)))                                     ; core.clj:113
